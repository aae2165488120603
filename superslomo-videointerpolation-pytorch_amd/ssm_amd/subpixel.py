"""conv3x3(upsample2x(cat[a, b])) in its sub-pixel form (profiles/DESIGN_history_r1-r3.md 3.1a): a plain 3x3 convolution of the LOW-res tensors with
4*Cout outputs - the effective filter M_a W M_b^T of each output parity (a, b) - and a pixel-shuffle store.  The bilinear rule
clamps at the border (scripts/models/flow_computation.py:215-289 via F.upsample, align_corners=False) while the convolution
zero-pads the upsampled map, so low-res rows / columns 0 and last have their own effective filters: border strips (rows as H = 1
views, columns through transposed copies with the transposed filter) and four corners.  The nine problems own disjoint output
pixels and run as ONE launch of the Q8 convolution kernel (ssm_conv16_subpixel_run); the filters are static (inference plans)."""
import ctypes

import torch

from . import hipbind as hb

# 1-D maps: M[class][parity][low-res tap t (i-1, i, i+1)][filter tap k] = weight of w_k on X[i + t - 1] in output 2i + parity
_M = {
    "int": [[[.75, .25, 0.], [.25, .75, .75], [0., 0., .25]],
            [[.25, 0., 0.], [.75, .75, .25], [0., .25, .75]]],
    "lo": [[[0., 0., 0.], [0., 1., .75], [0., 0., .25]],          # i = 0: U[-1] = 0 (conv padding), U[0] = X[0] (clamped rule)
           [[0., 0., 0.], [1., .75, .25], [0., .25, .75]]],
    "hi": [[[.75, .25, 0.], [.25, .75, 1.], [0., 0., 0.]],        # i = last: U[2H-1] = X[H-1], U[2H] = 0
           [[.25, 0., 0.], [.75, 1., 0.], [0., 0., 0.]]],
}


def effective_filter(w, cls_y, cls_x, transposed=False):
    """w [Co,Ci,3,3] -> [4*Co,Ci,3,3]: channel (2*pa + pb)*Co + c = filter of kernel-space parity (pa, pb).  transposed: the launch
    runs with its y axis along the image's columns (column strips): kernel parity pa / tap ty belong to the image's x axis."""
    My = torch.tensor(_M[cls_y], dtype=torch.float64, device=w.device)      # [a][ty][ky]
    Mx = torch.tensor(_M[cls_x], dtype=torch.float64, device=w.device)      # [b][tx][kx]
    e = torch.einsum("ayk,oikl,bxl->aboiyx", My, w.double(), Mx)            # [a][b][Co][Ci][ty][tx] in image axes
    if transposed:
        e = e.permute(1, 0, 2, 3, 5, 4)                                     # kernel (pa, pb, ty, tx) = image (b, a, tx, ty)
    co, ci = w.shape[:2]
    return e.reshape(4 * co, ci, 3, 3).to(torch.float32).contiguous()


class SubpixelUpConv:
    """One decoder level: a [B,Ca,h,w] and b [B,Cb,h,w] (Q8 HPlanes) -> dst [B,Co,2h,2w] (Q8 HPlanes)."""

    def __init__(self, weight, bias, Ga, Gb, B, h, w, device):
        assert h >= 2 and w >= 2, "the sub-pixel form needs at least 2x2 low-res pixels"
        self.B, self.h, self.w, self.Ga, self.Gb, self.device = B, h, w, Ga, Gb, device
        self.co = weight.shape[0]
        self.cin_p = 8 * (Ga + Gb)
        wt = weight.detach().to(device=device, dtype=torch.float32)
        if wt.shape[1] < self.cin_p:         # input channels padded like the source tensors' groups
            wt = torch.cat([wt, torch.zeros(wt.shape[0], self.cin_p - wt.shape[1], 3, 3, device=device)], 1)
        b4 = bias.detach().to(device=device, dtype=torch.float32).repeat(4)
        main = hb.PackedConv16(effective_filter(wt, "int", "int"), b4, w, q8=True)
        sc = main.scale

        def pk(cy, cx, transposed=False):    # every problem is packed for the main problem's tile configuration (width hint w)
            return hb.PackedConv16(effective_filter(wt, cy, cx, transposed), b4, w, q8=True, scale=sc)

        # (packed filter, source: None = a/b | "C" = the column copies, source pixel (y0, x0), extent (H, W), destination pixel,
        #  transposed, skip first/last row, skip first/last column of the extent)
        # the small problems first: their few workgroups (each a full pass over the input channels) start with the launch instead of
        # forming its tail
        self.problems = [
            (pk("lo", "lo"), None, (0, 0), (1, 1), (0, 0), False, False, False),
            (pk("lo", "hi"), None, (0, w - 1), (1, 1), (0, 2 * (w - 1)), False, False, False),
            (pk("hi", "lo"), None, (h - 1, 0), (1, 1), (2 * (h - 1), 0), False, False, False),
            (pk("hi", "hi"), None, (h - 1, w - 1), (1, 1), (2 * (h - 1), 2 * (w - 1)), False, False, False),
            (pk("lo", "int"), None, (0, 0), (1, w), (0, 0), False, False, True),
            (pk("hi", "int"), None, (h - 1, 0), (1, w), (2 * (h - 1), 0), False, False, True),
            (pk("int", "lo", True), "C", (0, 0), (1, h), (0, 0), True, False, True),
            (pk("int", "hi", True), "C", (4, 0), (1, h), (0, 2 * (w - 1)), True, False, True),
            (main, None, (0, 0), (h, w), (0, 0), False, True, True),
        ]
        G = Ga + Gb
        # image columns 0, 1 (rows 0, 1) and w-2, w-1 (rows 3, 4) as rows; row 2 stays zero
        self.cols = hb.HPlanes(B, 8 * G, 5, h, device, groups=G, q8=True)
        self._plan = None
        self.wcfg = w      # width hint of the tile configuration all nine problems share (the 32-wide 2-workgroup tile measured the same)

    def _build(self, a_view, b_view, dst, slope):
        lib = hb.load()
        n = len(self.problems)
        arr = (hb.SsmSubpixelProblem * n)()
        for q, (pk, src, (y0, x0), (H, W), (dy, dx), tr, sky, skx) in zip(arr, self.problems):
            if src is None:
                q.x1, q.C1 = a_view(y0, x0), 8 * self.Ga
                if self.Gb:
                    q.x2, q.C2 = b_view(y0, x0), 8 * self.Gb
                else:
                    q.x2, q.C2 = hb.NULL_HVIEW, 0
            else:
                q.x1, q.C1, q.x2, q.C2 = self.cols.view(y0=y0, x0=x0), 8 * (self.Ga + self.Gb), hb.NULL_HVIEW, 0
            q.w_packed, q.bias_packed, q.inv_wscale = pk.w.data_ptr(), pk.b.data_ptr(), 1.0 / pk.scale
            q.y_hl8 = dst.view(y0=dy, x0=dx)
            q.H, q.W, q.transposed, q.skip_y, q.skip_x = H, W, 1 if tr else 0, 1 if sky else 0, 1 if skx else 0
        nbytes = lib.ssm_conv16_subpixel_table_bytes(n)
        host = (ctypes.c_char * nbytes)()
        starts = (ctypes.c_int * (n + 1))()
        flags = hb.SSM_FLAG_LRELU | hb.SSM_FLAG_Q8
        hb.check(lib.ssm_conv16_subpixel_plan(ctypes.cast(arr, ctypes.c_void_p), n, self.B, self.co, self.wcfg, slope, flags,
                                              ctypes.cast(host, ctypes.c_void_p), nbytes, starts))
        table = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(self.device)
        return table, starts, n

    def run(self, a_view, b_view, dst, slope=0.1):
        """a_view / b_view: callables (y0, x0) -> ssm_hview of the sources at that pixel; dst: HPlanes [B,Co,2h,2w] (Q8)."""
        lib, st = hb.load(), hb.stream_ptr()
        a0 = a_view(0, 0)
        b0 = b_view(0, 0) if self.Gb else hb.NULL_HVIEW
        key = (a0.ptr, b0.ptr, dst.buf.data_ptr(), slope)
        if self._plan is None or self._plan[0] != key:       # the table holds raw pointers of the views
            self._plan = (key,) + self._build(a_view, b_view, dst, slope)
        _, table, starts, n = self._plan
        hb.check(lib.ssm_hl8_gather_cols(a0, self.Ga, b0, self.Gb, self.cols.view(), self.B, self.h, 0, 2, self.w - 2, st))
        hb.check(lib.ssm_conv16_subpixel_run(table.data_ptr(), n, starts, self.co, self.wcfg, self.cin_p, st))
