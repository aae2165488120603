"""The plain-C oracle (oracle/ssm_oracle.c) against the reference's golden vectors and against the
torch oracle - two independent restatements must agree.  No GPU."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import ssm_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "libssm_oracle.so")
fp = ctypes.POINTER(ctypes.c_float)


@pytest.fixture(scope="module")
def C():
    if not os.path.exists(SO):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    return ctypes.CDLL(SO)


def P(a):
    return a.ctypes.data_as(fp)


def F(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def test_c_conv(C, golden):
    g = golden("ops")
    for tag in ("conv_k7_c6_n32", "conv_k5_c32_n64", "conv_k3_c64_n32", "conv_k3_c32_n5"):
        x, w, b = F(g[tag + "_x"]), F(g[tag + "_w"]), F(g[tag + "_b"])
        y = np.empty_like(F(g[tag + "_y"]))
        Bn, Cin, H, W = x.shape
        C.oracle_conv2d(P(x), P(w), P(b), P(y), Bn, Cin, H, W, w.shape[0], w.shape[2], 1, ctypes.c_float(0.1))
        assert np.abs(y - g[tag + "_y"]).max() < 2e-5
        C.oracle_conv2d(P(x), P(w), P(b), P(y), Bn, Cin, H, W, w.shape[0], w.shape[2], 0, ctypes.c_float(0.1))
        assert np.abs(y - g[tag + "_ylin"]).max() < 2e-5


def test_c_pool_upsample_warp(C, golden):
    g = golden("ops")
    x = F(g["pool_x"])
    y = np.empty_like(F(g["pool_y"]))
    C.oracle_avgpool2(P(x), P(y), x.shape[0] * x.shape[1], x.shape[2], x.shape[3])
    assert np.abs(y - g["pool_y"]).max() < 1e-6
    cat = F(np.concatenate([g["up_a"], g["up_b"]], 1))
    up = np.empty_like(F(g["up_y"]))
    C.oracle_upsample2x(P(cat), P(up), cat.shape[0] * cat.shape[1], cat.shape[2], cat.shape[3])
    assert np.abs(up - g["up_y"]).max() < 1e-6
    img, flo = F(g["warp_img"]), F(g["warp_flo"])
    out = np.empty_like(img)
    C.oracle_warp(P(img), P(flo), P(out), *img.shape)
    assert np.abs(out - g["warp_y"]).max() < 1e-5


def test_c_inputs_synthesis(C, golden):
    g = golden("ops")
    img6, flow4, out5 = F(g["fi_img6"]), F(g["fi_flow4"]), F(g["fi_out5"])
    B, _, H, W = img6.shape
    for i, tv in enumerate((0.125, 0.5, 0.875)):
        t = np.full(B, tv, dtype=np.float32)
        in16 = np.empty((B, 16, H, W), dtype=np.float32)
        C.oracle_flowinterp_inputs(P(img6), P(flow4), P(t), P(in16), B, H, W)
        assert np.abs(in16 - g["fi_in16_%d" % i]).max() < 1e-5
        y3 = np.empty((B, 3, H, W), dtype=np.float32)
        C.oracle_synthesize(P(img6), P(F(g["fi_in16_%d" % i])), P(out5), P(t), P(y3), B, H, W)
        assert np.abs(y3 - g["fi_img_%d" % i]).max() < 5e-5


def test_c_vs_torch_oracle_random(C):
    rng = np.random.RandomState(3)
    img = F(rng.randn(1, 3, 13, 21))
    flo = F(rng.randn(1, 2, 13, 21) * 6)
    out = np.empty_like(img)
    C.oracle_warp(P(img), P(flo), P(out), 1, 3, 13, 21)
    assert np.abs(out - O.warp(torch.from_numpy(img), torch.from_numpy(flo)).numpy()).max() < 1e-5
