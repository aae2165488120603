"""Bitwise determinism of a PairEngine: the same pair run (a) repeatedly on one stream and (b) on two engines on two
concurrent HIP streams must leave identical bytes in every activation buffer.  Prints, per buffer in execution order,
how many runs differed from the first.  Usage: python tools/determinism_probe.py [mode] [H] [W] [reps] [mode of the 2nd stream's engine]
$SSM_PROBE_LIB: load this build of libssm_hip.so instead (e.g. one compiled with the SLP vectoriser on, to reproduce the hazard)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd")
for p in (ROOT, PKG, os.path.join(PKG, "scripts")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from ssm_amd import hipbind as _hb  # noqa: E402

if os.environ.get("SSM_PROBE_LIB"):
    _hb.LIB_PATH = os.path.abspath(os.environ["SSM_PROBE_LIB"])
from ssm_amd.engine import PairEngine  # noqa: E402
from ssm_amd.weights import synthetic_frames, synthetic_state_dict  # noqa: E402


def snapshot(eng):
    out = {}
    for tag, plan in (("s1", eng.s1), ("s2", eng.s2)):
        for name, pl in plan.t.items():
            out["%s.%s" % (tag, name)] = pl.buf.clone()
    out["img"] = eng.img.clone()
    if eng.est is not None:
        out["est"] = eng.est.clone()
    return out


def plan_items(eng):
    for tag, plan in (("s1", eng.s1), ("s2", eng.s2)):
        for name, pl in plan.t.items():
            if hasattr(pl, "Hp") and hasattr(pl, "G"):
                yield "%s.%s" % (tag, name), pl


def first_of(names, bad):
    for n in names:
        if bad[n]:
            return n
    return None


def order(eng):
    names = []
    for tag, plan in (("s1", eng.s1), ("s2", eng.s2)):
        names += ["%s.in" % tag] + ["%s.%s" % (tag, n) for n in plan.CONV_OUT if n in plan.t]
        names += ["%s.%s" % (tag, n) for n in plan.t if "%s.%s" % (tag, n) not in names]
    return names + ["img"] + (["est"] if eng.est is not None else [])


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "f16f8"
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 96
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    mode2 = sys.argv[5] if len(sys.argv) > 5 else mode       # what the OTHER stream runs beside engine 0
    dev = torch.device("cuda:0")
    sd1, sd2 = synthetic_state_dict(1), synthetic_state_dict(2)
    sd1 = {k: v.to(dev) for k, v in sd1.items()}
    sd2 = {k: v.to(dev) for k, v in sd2.items()}
    x = synthetic_frames(2, H, W, seed=20).to(dev).reshape(1, 6, H, W)
    t = torch.tensor([0.25, 0.5, 0.75], device=dev)
    e0 = PairEngine(sd1, sd2, 1, 3, H, W, dev, True, mode)
    e1 = PairEngine(sd1, sd2, 1, 3, H, W, dev, True, mode2)
    torch.cuda.synchronize()
    e0.run(x, t, False)
    torch.cuda.synchronize()
    ref = snapshot(e0)
    names = order(e0)
    for label in ("one stream", "two streams"):
        bad = {n: 0 for n in names}
        worst = {n: 0.0 for n in names}
        for _ in range(reps):
            if label == "one stream":
                e0.run(x, t, False)
                torch.cuda.synchronize()
                snaps = [snapshot(e0)]
            else:
                sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
                sa.wait_stream(torch.cuda.current_stream())
                sb.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(sa):
                    e0.run(x, t, False)
                with torch.cuda.stream(sb):
                    e1.run(x, t, False)
                torch.cuda.synchronize()
                snaps = [snapshot(e0)] + ([snapshot(e1)] if mode2 == mode else [])
            for s in snaps:
                for n in names:
                    bits = (lambda v: v.view(torch.int16) if v.dtype == torch.float16 else v.view(torch.int32))
                    if not torch.equal(bits(s[n]), bits(ref[n])):
                        bad[n] += 1
                        a, b = s[n].float(), ref[n].float()
                        worst[n] = max(worst[n], float((a - b).abs().nan_to_num(1e9).max()))
                        if n == "est":
                            d = (s[n] != ref[n]).nonzero()
                            print("  [est] %d differing floats; channels %s; rows %s; cols %d..%d" % (
                                d.shape[0], sorted(set(d[:, 1].tolist())), sorted(set(d[:, 2].tolist())), int(d[:, 3].min()), int(d[:, 3].max())))
                            flow = e0.s1.t["out"].interior
                            for i in d[:4].tolist():
                                b_, _, y_, x_ = i
                                tt = float(t[b_])
                                c00 = torch.tensor(-(1.0 - tt), dtype=torch.float32) * torch.tensor(tt, dtype=torch.float32)
                                c01 = torch.tensor(tt, dtype=torch.float32) ** 2
                                f01u, f10u = flow[0, 0, y_, x_].cpu(), flow[0, 2, y_, x_].cpu()
                                ft0v = s[n][b_, 3, y_, x_].cpu()
                                hyp = c01 * f10u + ft0v
                                print("     b %d c %d y %d x %d: got %.7f ref %.7f | c01*f10u + ft0v = %.7f  (c00*f01u = %.7f, ft0v = %.7f)" % (
                                    i[0], i[1], i[2], i[3], float(s[n][tuple(i)]), float(ref[n][tuple(i)]), float(hyp), float(c00 * f01u), float(ft0v)))
                        if n == first_of(names, bad) and bad[n] <= 3 and n in dict(plan_items(e0)):
                            pl = dict(plan_items(e0))[n]
                            idx = (bits(s[n]) != bits(ref[n])).nonzero().flatten()
                            pix = pl.Hp * pl.Wp * 8
                            print("  [%s] %d differing halfwords, first %d last %d; Hp %d Wp %d G %d" % (n, idx.numel(), int(idx[0]), int(idx[-1]), pl.Hp, pl.Wp, pl.G))
                            for i in idx[:6].tolist() + idx[-3:].tolist():
                                bg, r = divmod(i, 2 * pix)
                                plane, r = divmod(r, pix)
                                yy, r = divmod(r, pl.Wp * 8)
                                xx, e = divmod(r, 8)
                                print("     b*G+g %d plane %d y %d x %d e %d: got %04x ref %04x" % (bg, plane, yy, xx, e, int(bits(s[n])[i]) & 0xffff, int(bits(ref[n])[i]) & 0xffff))
        print("== %s (%s%s, %dx%d, %d reps)" % (label, mode, "" if label == "one stream" else " beside " + mode2, H, W, reps))
        for n in names:
            if bad[n]:
                print("  %-10s differed in %d snapshots, max |diff| %.3e" % (n, bad[n], worst[n]))
        if not any(bad.values()):
            print("  all buffers bit-identical")


if __name__ == "__main__":
    main()
