"""The C ABI's threading contract (include/ssm_hip.h "Conventions"; SURVEY 8b threading row): every entry point launches on the calling
thread's current device and stream and is re-entrant.  The reference's callers are torch.nn.DataParallel replica threads
(scripts/main.py:74-76: one Python thread per device inside one process), so every kernel family that opts into more than 64 KiB of
dynamic LDS - the opt-in is per (kernel, device), csrc/ssm_common.h reserve_lds - is launched here from TWO host threads on TWO streams at
once (ctypes drops the GIL inside the call) and must give the bits of the single-threaded launch.  One-GPU boxes: both threads drive
cuda:0; the (kernel, device) keying itself is fenced on the CPU by tests/test_build_fences_cpu.py."""
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


def _families(hb):
    """(name, packer, launcher, k, cin, cout, B, H, W): one problem per > 64 KiB kernel family."""
    return [
        ("wino2 F(2x2,3x3)", hb.PackedWino, hb.conv2d_wino, 3, 64, 64, 2, 22, 40),
        ("wino4 F(4x4,3x3) 64-cout", hb.PackedWino4, hb.conv2d_wino4, 3, 64, 64, 2, 32, 64),
        ("wino4 F(4x4,3x3) 32-cout", hb.PackedWino4, hb.conv2d_wino4, 3, 32, 32, 2, 32, 64),
        ("wino5s F(4x4,5x5)", hb.PackedWino5, hb.conv2d_wino5, 5, 32, 64, 2, 32, 64),
        ("wino7s blocked 7x7", hb.PackedWino7, hb.conv2d_wino7, 7, 8, 32, 2, 32, 64),
        ("wino1d F(2,7)", hb.PackedWino1d, hb.conv2d_wino1d, 7, 8, 32, 2, 32, 64),
        ("wino1d F(4,5)", hb.PackedWino1d, hb.conv2d_wino1d, 5, 32, 64, 2, 32, 64),
    ]


def test_every_large_lds_family_from_two_threads_on_two_streams(dev):
    from ssm_amd import hipbind as hb
    g = torch.Generator().manual_seed(11)
    for name, Packer, launch, k, cin, cout, B, H, W in _families(hb):
        x = torch.randn(B, cin, H, W, generator=g)
        w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
        bias = torch.randn(cout, generator=g) * 0.1
        pk = Packer(w.to(dev), bias.to(dev), B, H, W)
        px = hb.Planes(B, pk.cin_p, H, W, dev)
        px.interior[:, :cin] = x.to(dev)
        y0 = hb.Planes(B, cout, H, W, dev)
        launch(px.view(), pk.cin_p, None, 0, pk, y0.view(), None, B, H, W, lrelu=True)
        torch.cuda.synchronize()
        want = y0.full.clone()
        assert float(want.abs().max()) > 0.1, name

        outs = [[hb.Planes(B, cout, H, W, dev) for _ in range(4)] for _ in range(2)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        errors, start = [], threading.Barrier(2)

        def worker(i):
            try:
                torch.cuda.set_device(dev)                      # what a DataParallel replica thread does first
                with torch.cuda.stream(streams[i]):
                    assert hb.stream_ptr().value == streams[i].cuda_stream
                    start.wait()
                    for rep in range(24):
                        launch(px.view(), pk.cin_p, None, 0, pk, outs[i][rep % 4].view(), None, B, H, W, lrelu=True)
                streams[i].synchronize()
            except Exception as e:                              # noqa: BLE001 - reported by the asserting thread
                errors.append((i, repr(e)))

        torch.cuda.synchronize()
        threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, "%s: %s" % (name, errors)
        for i in range(2):
            for o in outs[i]:
                assert torch.equal(o.full, want), "%s: thread %d's launch differs from the single-threaded one" % (name, i)


def test_error_strings_are_per_thread(dev):
    """ssm_last_error_string() is thread-local: a refused call on one thread does not show on another."""
    from ssm_amd import hipbind as hb
    lib = hb.load()
    seen = {}

    def bad():
        x = hb.Planes(1, 8, 8, 8, dev)
        rc = lib.ssm_avgpool2_fwd(x.view(), x.view(), 1, 8, 7, 7, hb.stream_ptr())        # odd sizes: refused
        seen["bad"] = (rc, lib.ssm_last_error_string())

    def good():
        seen["good"] = lib.ssm_last_error_string()

    t = threading.Thread(target=bad)
    t.start()
    t.join()
    t = threading.Thread(target=good)
    t.start()
    t.join()
    assert seen["bad"][0] != 0 and seen["bad"][1]
    assert not seen["good"]
