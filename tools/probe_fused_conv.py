import torch, time, sys
sys.path.insert(0, '.')
from adverse_weather_semantic_segmentation_robustness_benchmark_amd import ops
torch.manual_seed(0)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(True); e=torch.cuda.Event(True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n
for (cin,cout,k,H,W) in [(256,64,1,256,512),(64,64,3,256,512),(64,256,1,256,512),(1024,256,1,64,128),(512,512,3,64,128)]:
    x=torch.randn(8,cin,H,W,device='cuda').contiguous(memory_format=torch.channels_last)
    w=torch.randn(cout,cin,k,k,device='cuda').contiguous(memory_format=torch.channels_last)*0.05
    b=torch.randn(cout,device='cuda')
    pad=k//2
    def a():
        y=torch.nn.functional.conv2d(x,w,None,1,pad)
        ops.bias_act_nhwc_(y.permute(0,2,3,1),b,None,1); return y
    def c():
        return torch.miopen_convolution_relu(x,w,b,[1,1],[pad,pad],[1,1],1)
    def d():
        return torch.nn.functional.conv2d(x,w,None,1,pad)
    try:
        ya=a(); yc=c()
        err=(ya-yc).abs().max().item()
        print(cin,cout,k,H,W,'conv only %.3f  conv+epi %.3f  miopen_conv_relu %.3f  err %.2e  cl=%s'%(t(d),t(a),t(c),err,yc.is_contiguous(memory_format=torch.channels_last)))
    except Exception as ex:
        print('fail',cin,cout,k,repr(ex)[:200])
# layer norm probe
for (n,c) in [(8*256*512,32),(8*128*256,64),(8*64*128,160),(8*32*64,256)]:
    x=torch.randn(n,c,device='cuda'); wt=torch.randn(c,device='cuda'); bs=torch.randn(c,device='cuda')
    print('LN',n,c,'%.3f ms'%t(lambda: torch.nn.functional.layer_norm(x,(c,),wt,bs,1e-5)), 'ideal %.3f'%(2*n*c*4/5e9))
