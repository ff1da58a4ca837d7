"""CPU check of tests/forms_ref.py: the float64 derivation of the bilinear forms behind awseg_upconv_forms /
awseg_depth_head_fused equals relu(shift + conv3x3(F.interpolate(x32, bilinear, align_corners=False))) on every pixel of every
16 x 16 tile patch, half cells and image borders included (the -m gpu tests then pin the HIP tables to this derivation)."""
import numpy as np

from tests import forms_ref


def test_forms_equal_interpolate_then_conv():
    rs = np.random.RandomState(3)
    for (h, w) in [(1, 1), (1, 2), (2, 1), (2, 3)]:
        C = 3
        G = rs.randn(h, w, 9, C)
        shift = rs.randn(C)
        ref = forms_ref.reference(G, shift)
        F4, F2 = forms_ref.build_tables(G, shift)
        H, W = 32 * h, 32 * w
        refp = np.zeros((H + 2, W + 2, C))
        refp[1:-1, 1:-1] = ref
        for my in range(H // 16):
            for mx in range(W // 16):
                p = forms_ref.generate_tile(F4, F2, h, w, my, mx)
                assert np.abs(p - refp[16 * my:16 * my + 18, 16 * mx:16 * mx + 18]).max() < 1e-12
