import sys, time, torch
sys.path.insert(0, '.')
import adverse_weather_semantic_segmentation_robustness_benchmark_amd as P
from adverse_weather_semantic_segmentation_robustness_benchmark_amd.data.loader import CityscapesKITTIDataset
from adverse_weather_semantic_segmentation_robustness_benchmark_amd.evaluation.harness import EvalState, AUROC_LO, AUROC_HI
from adverse_weather_semantic_segmentation_robustness_benchmark_amd import ops, _native as N
import torch.nn.functional as F
torch.manual_seed(0)
model = P.EnsembleModel(pretrained=False).cuda().eval()
ds = CityscapesKITTIDataset(split="test", image_size=(1024, 2048), weather_schedule="round_robin", num_samples=64)
st = EvalState(P.RobustnessMetrics(19), ds.weather_conditions, "cuda", 15, True)
def sync(): torch.cuda.synchronize(); return time.time()
with torch.no_grad():
    for it in range(4):
        t0 = sync(); batch = ds.make_batch(it * 8, 8); t1 = sync()
        cond = st.acc.cond_ids(batch["weather_condition"])
        res = model.forward_eval(batch["image"], batch["label"], st.acc.counts, st.acc.oob, cond, want_logits=False, want_pred=False); t2 = sync()
        w = F.softmax(model.ensemble_weights, dim=0)
        ops.ensemble_eval_stats(res["segformer_seg"], res["deeplabv3plus_seg"], 0, w, model.temperature, batch["label"], cond, st.edges, st.ece, st.auroc, AUROC_LO, AUROC_HI); t3 = sync()
        print("batch %d: make_batch %.1f ms, forward_eval %.1f ms, stats %.1f ms" % (it, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), flush=True)
    # host-only cost of make_batch pieces
    t0 = time.time(); conds = ds.choose_conditions(0, 8); imgs, labels = ds.synth_raw(8); sync(); t1 = time.time()
    print("synth_raw %.1f ms" % ((t1 - t0) * 1e3))
