"""The hard-edge parity bars of tests/test_hip_model.py are written in units of d = max|oracle fp32 - oracle float64| on four
deterministic (weight family, frame family) inputs, and d is a PINNED constant there (ORACLE_OWN_ROUNDING_720P: 70 s of float64 per
family).  A pinned constant is only as good as the inputs it was measured on: this CPU test pins the generators themselves, so a change
to ssm_amd.weights.synthetic_frames_u8 / synthetic_frames_edges_u8 / synthetic_state_dict fails HERE and tells the author to re-measure d
(`python tests/study_parity_families.py --f64`, or read parity.families[*].oracle_fp32_vs_f64 of a `python bench.py` line) instead of
leaving the constants stale."""
import zlib

import pytest
import torch

# uint8 frames at the size the constants were measured on: exact
FRAME_CRC32 = {"texture": 1343641871, "edges": 696213158}
# (count, sum, sum |w|, sum w^2) of every tensor of a stage's state dict in float64: tolerant of a last-bit difference between hosts
WEIGHT_STATS = {
    ("uniform", 1): (18236644, 10.539041149873789, 353455.0569426688, 9982.155835508225),
    ("uniform", 2): (20611909, 228.7791969809, 371549.3879110855, 9995.27298536212),
    ("smooth", 1): (18236644, 43.36996532350621, 359915.02008482744, 12226.998277728675),
    ("smooth", 2): (20611909, -40.96185805235974, 380510.04353006487, 12239.585458816697),
}


def test_frame_generators_behind_the_pinned_rounding_constants():
    from ssm_amd.weights import synthetic_frames_edges_u8, synthetic_frames_u8
    tex = synthetic_frames_u8(2, 720, 1280, 42)
    edg = synthetic_frames_edges_u8(2, 720, 1280, 7)
    assert tex.dtype == torch.uint8 and edg.dtype == torch.uint8
    got = {"texture": zlib.crc32(tex.contiguous().numpy().tobytes()), "edges": zlib.crc32(edg.contiguous().numpy().tobytes())}
    assert got == FRAME_CRC32, ("the synthetic frame generators changed: re-measure ORACLE_OWN_ROUNDING_720P (tests/test_hip_model.py) and "
                                "update FRAME_CRC32: %s" % got)


@pytest.mark.parametrize("family,stage", sorted(WEIGHT_STATS))
def test_weight_families_behind_the_pinned_rounding_constants(family, stage):
    from ssm_amd.weights import synthetic_state_dict
    sd = synthetic_state_dict(stage, family=family)
    v = torch.cat([sd[k].double().flatten() for k in sorted(sd)])
    n, s1, sa, s2 = WEIGHT_STATS[(family, stage)]
    assert v.numel() == n
    assert abs(float(v.abs().sum()) - sa) < 1e-7 * sa and abs(float((v * v).sum()) - s2) < 1e-7 * s2 and abs(float(v.sum()) - s1) < 1e-6 * sa / n ** 0.5 + 1e-6, \
        "weight family %s / stage %d changed: re-measure ORACLE_OWN_ROUNDING_720P (tests/test_hip_model.py)" % (family, stage)


def test_the_constants_are_keyed_on_these_families():
    import ast
    import os
    src = open(os.path.join(os.path.dirname(__file__), "test_hip_model.py")).read()
    line = next(ln for ln in src.splitlines() if ln.startswith("ORACLE_OWN_ROUNDING_720P"))
    table = ast.literal_eval(line.split("=", 1)[1].strip())
    assert sorted(table) == sorted((w, f) for w in ("uniform", "smooth") for f in ("texture", "edges"))
    assert all(1e-4 < d < 3e-3 for d in table.values())
