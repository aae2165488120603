"""CPU checks of bench.py's FLOP accounting (no GPU): the algorithmic conv FLOP of a pair must be SURVEY 8d's figure, and the
"issued" figures of the fp32 modes (what the matrix cores execute after the per-pair hoisting and the Winograd form) must follow from
the layer table."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_algorithmic_flop_per_pair_is_the_survey_figure():
    b = _bench()
    fl = b.conv_flops_per_pair(736, 1280, 7)
    assert abs(fl / 1e12 - 5.855) < 1e-3           # 345.18 + 7 x 368.90 GMAC = 5.855 TFLOP


def test_issued_flop_follows_from_hoisting_and_winograd():
    b = _bench()
    from ssm_amd.weights import unet_layers
    alg = b.conv_flops_per_pair(736, 1280, 7)
    f32 = b.conv_flops_per_pair(736, 1280, 7, "f32")
    f32w = b.conv_flops_per_pair(736, 1280, 7, "f32w")
    px = 736 * 1280
    c1a = 2.0 * px * 16 * 32 * 49                  # stage-2 conv1a per t
    c7a = 2.0 * (px / 256) * 1024 * 512 * 9        # stage-2 conv7a per t (1/16 resolution)
    saved = 6 * (6 / 16) * c1a + 6 * 0.5 * c7a     # six of seven evaluations of the t-independent channels
    assert abs((alg - f32) - saved) / saved < 1e-9
    # mode f32w: every 3x3 layer except the final convs as F(4x4,3x3) (36 products per 16 outputs instead of 144) - or as F(2x2,3x3)
    # (16 per 4 outputs instead of 36) on the 23x40 maps (csrc/ssm_wino4.hip, ssm_wino4_preferred); the 7x7 layers as 2x2 blocks of F(4x4,4x4) (4 x 49 products per 16 outputs instead of 784: csrc/ssm_wino7.hip), the 5x5
    # layers as two-dimensional F(4x4,5x5) (64 products per 16 outputs instead of 400: csrc/ssm_wino5.hip)
    scale = {"conv1": 1, "conv2": 2, "conv3": 4, "conv4": 8, "conv5": 16, "conv6": 32, "conv7": 16, "conv8": 8, "conv9": 4, "fuse_": 1, "final": 1}
    want = 0.0
    for st, reps in ((1, 1), (2, 7)):
        for name, cin, cout, k in unet_layers(st, True):
            s = 2 if name.startswith("conv10") else 1 if name.startswith("conv11") else [v for p, v in scale.items() if name.startswith(p)][0]
            fl = 2.0 * (736 // s) * (1280 // s) * cin * cout * k * k * reps
            if st == 2 and name == "conv1a":
                fl *= (10 * 7 + 6) / (16 * 7)
            if st == 2 and name == "conv7a":
                fl *= 8 / 14
            if k == 3 and name != "final_conv":
                f22 = (736 // s) * (1280 // s) < 2048
                fl *= 16 / 36 if f22 else 36 / 144
            elif k == 7:
                fl *= 196 / 784
            elif k == 5:
                fl *= 64 / 400
            want += fl
    assert abs(f32w - want) / want < 1e-9
    assert f32w < f32 < alg
