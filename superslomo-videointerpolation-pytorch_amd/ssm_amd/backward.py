"""Backward pass of the hot path on the HIP kernels (training step, SURVEY 8f-1 / BASELINE config 3).

The reference trains through autograd over stock PyTorch ops; here every adjoint is a C-ABI call on fp32 padded
planes (the training plan runs in precision mode "f32"):
  conv data gradient   = the forward convolution kernel on the transposed, spatially flipped filter
  conv weight gradient = ssm_conv2d_wgrad, bias gradient = ssm_bias_grad
  LeakyReLU' + adjoint of the fused 2x2 mean = ssm_lrelu_bwd
  adjoint of concat + bilinear x2 = ssm_upsample2x_cat_bwd
  adjoints of compute_inputs / compute_output_image fused with the L1 loss gradients = ssm_flowinterp_inputs_bwd /
  ssm_synthesize_bwd
"""
import contextlib
import math
import os

import torch

from . import hipbind as hb


DGRAD_BLOCKED = os.environ.get("SSM_TRAIN_DGRAD_BLOCKED", "1") != "0"
DGRAD_WINO4 = os.environ.get("SSM_TRAIN_DGRAD_WINO4", "1") != "0"          # 3x3 data gradients as F(4x4,3x3) where the cost model prefers it
# r6: weight gradients of the 3x3 layers of an f32w plan in the Winograd domain (csrc/ssm_wgradw.hip: 16 instead of 36 multiplies per 2x2
# tile, like that plan's forward and data gradients) on the maps hb.wgrad_wino_supported accepts; $SSM_WGRAD_WINO=0: the direct kernel
WGRAD_WINO = os.environ.get("SSM_WGRAD_WINO", "1") != "0"
WGRAD_WINO_ALL = os.environ.get("SSM_WGRAD_WINO", "1") == "all"
# r6: LeakyReLU' of the layer BELOW fused into the kernel that produces its upstream gradient - the F(2x2) / F(4x4) data-gradient launch of
# the layer above (SSM_FLAG_MASK) or the upsample adjoint - where that gradient has one source; $SSM_LRELU_FUSE=0: a separate ssm_lrelu_bwd
# launch per layer as before
LRELU_FUSE = os.environ.get("SSM_LRELU_FUSE", "1") != "0"
# data-gradient forms that write dZ of the layer below through the mask epilogue.  The 5x5 layers' F(4x4,5x5) form has the epilogue too
# (ssm_wino5_conv2d_add_fwd, bit-identical: tests/test_hip_backward.py) but the step does not use it by default ($SSM_LRELU_FUSE_W5=1: on):
# its epilogue reads the mask source without the prefetch the 3x3 forms have and the launch grows by what the ssm_lrelu_bwd pass took
# (wino5s 0.53 -> 0.83 ms per step, lrelu_bwd 0.86 -> 0.56; step 149.2 / 149.3 / 149.2 off, 149.3 / 147.2 / 149.0 on: profiles/r54_w5fuse.txt)
FUSE_ALGOS = ("wino", "wino4", "wino5") if os.environ.get("SSM_LRELU_FUSE_W5", "0") != "0" else ("wino", "wino4")


def transposed_filter(w):
    """OIHW filter of the data-gradient convolution: W'[ci][co][ky][kx] = W[co][ci][k-1-ky][k-1-kx]."""
    return w.detach().permute(1, 0, 2, 3).flip(2, 3).contiguous()


def lrelu_bwd(dy, dpool, y, dz, slope=0.1, has_act=True):
    """dz = (dy + 1/4 dpool^) * LeakyReLU'(y).  dy / dpool / y / dz: hb.Planes (dy or dpool may be None)."""
    hb.check(hb.load().ssm_lrelu_bwd(dy.view() if dy is not None else hb.NULL_VIEW,
                                     dpool.view() if dpool is not None else hb.NULL_VIEW,
                                     y.view() if y is not None else hb.NULL_VIEW, dz.view(), dz.B, dz.C, dz.H, dz.W, slope,
                                     1 if has_act else 0, hb.stream_ptr()))
    return dz


def bias_grad(dz, out, zero_first=True):
    fn = hb.load().ssm_bias_grad if zero_first else hb.load().ssm_bias_grad_acc
    hb.check(fn(dz.view(), out.data_ptr(), dz.B, dz.C, dz.H, dz.W, hb.stream_ptr()))
    return out


def wgrad(x, dz, out, k, ci_offset=0, zero_first=True, split=False, bias_acc=None):
    """out[:, ci_offset:ci_offset+x.C] (OIHW fp32, contiguous) = sum_{b,y,x} dz * x(shifted).
    x, dz: hb.Planes (or slices) of the layer's input / dZ; two-source convs call this once per source.
    split: the bf16 matrix path with hi/lo-split operands (ssm_conv2d_wgrad_bf16x3) instead of the fp32 one.
    bias_acc (fp32 path): [Cout] tensor that ALSO receives += sum dz - the bias gradient as one more column of the same GEMM
    (ssm_conv2d_wgrad_bias); pass it with one source of a two-source layer."""
    assert out.is_contiguous() and out.shape[0] == dz.C and out.shape[2] == k and ci_offset + x.C <= out.shape[1]
    lib = hb.load()
    if bias_acc is not None:
        assert not split and bias_acc.is_contiguous() and bias_acc.numel() == dz.C and bias_acc.dtype == torch.float32
        hb.check(lib.ssm_conv2d_wgrad_bias(x.view(), dz.view(), out.data_ptr(), bias_acc.data_ptr(), x.B, x.C, dz.C, x.H, x.W, k,
                                           out.shape[1], ci_offset, 1 if zero_first else 0, hb.stream_ptr()))
        return out
    fn = lib.ssm_conv2d_wgrad_bf16x3 if split else lib.ssm_conv2d_wgrad
    hb.check(fn(x.view(), dz.view(), out.data_ptr(), x.B, x.C, dz.C, x.H, x.W, k, out.shape[1],
                ci_offset, 1 if zero_first else 0, hb.stream_ptr()))
    return out


def upsample_cat_bwd(du, da, db=None, acc_a=False, acc_b=False, mask_a=None, slope=0.1):
    """mask_a (Planes: the output of the layer that produced the a-source): da receives dZ of that layer, the adjoint x LeakyReLU'(mask_a)."""
    if mask_a is not None:
        hb.check(hb.load().ssm_upsample2x_cat_bwd_mask(du.view(), da.view(), da.C, db.view() if db is not None else hb.NULL_VIEW,
                                                       db.C if db is not None else 0, mask_a.view(), slope, da.B, da.H, da.W,
                                                       1 if acc_a else 0, 1 if acc_b else 0, hb.stream_ptr()))
        return
    hb.check(hb.load().ssm_upsample2x_cat_bwd(du.view(), da.view(), da.C, db.view() if db is not None else hb.NULL_VIEW,
                                              db.C if db is not None else 0, da.B, da.H, da.W, 1 if acc_a else 0,
                                              1 if acc_b else 0, hb.stream_ptr()))


# ------------------------------------------------------------------------------------------------------------------
# U-Net backward over a UNetPlan (mode "f32", concat+upsample materialised): every activation the adjoints need is
# still in the plan's buffers (nothing is aliased), so the backward is a second fixed launch sequence.
# ------------------------------------------------------------------------------------------------------------------
from .engine import POOLED, layer_scale  # noqa: E402
from .weights import param_key  # noqa: E402

# (layer, input tensor(s), output tensor) in forward order - scripts/models/flow_computation.py:155-289
_CONV_IO = [
    ("conv1a", ("in",), "t1a"), ("conv1b", ("t1a",), "c1"), ("conv2a", ("p2",), "t2a"), ("conv2b", ("t2a",), "c2"),
    ("conv3a", ("p3",), "t3a"), ("conv3b", ("t3a",), "c3"), ("conv4a", ("p4",), "t4a"), ("conv4b", ("t4a",), "c4"),
    ("conv5a", ("p5",), "t5a"), ("conv5b", ("t5a",), "c5"), ("conv6.0", ("p6",), "t6a"), ("conv6.1", ("t6a",), "c6"),
    ("conv7a", ("u7",), "t7a"), ("conv7b", ("t7a",), "c7"), ("conv8a", ("u8",), "t8a"), ("conv8b", ("t8a",), "c8"),
    ("conv9a", ("u9",), "t9a"), ("conv9b", ("t9a",), "c9"), ("conv10a", ("u10",), "t10a"), ("conv10b", ("t10a",), "c10"),
    ("conv11a", ("u11",), "t11a"), ("conv11b", ("t11a",), "c11"), ("fuse_conv", ("c11", "c1"), "tf"),
    ("final_conv", ("tf",), "out"),
]
_POOL_OF = {"conv1b": "p2", "conv2b": "p3", "conv3b": "p4", "conv4b": "p5", "conv5b": "p6"}
_UP_SOURCES = {"u7": ("c6", None), "u8": ("c7", "c5"), "u9": ("c8", "c4"), "u10": ("c9", "c3"), "u11": ("c10", "c2")}


def grad_layout(layers, n_buckets):
    """Layout of a U-Net's parameter gradients in ONE flat fp32 buffer, a pure function of the layer table {name: (cin, cout, k)} in
    state-dict order: ([(state-dict key, shape)], {layer: [a, b) of its weight + bias}, buckets).  The backward walks the layers in
    reverse state-dict order, so the buffer completes from its tail: a bucket = a run of consecutive layers of roughly 1 / n_buckets
    of the floats.  Shared by UNetGrad and by `bench.py --mode train --stub` (the gloo rehearsal of the exchange at world 8)."""
    sizes = []
    for name, (ci, co, k) in layers.items():
        sizes += [(param_key(name, "weight"), (co, ci, k, k)), (param_key(name, "bias"), (co,))]
    off, span = 0, {}
    for key, sh in sizes:
        n = 1
        for d in sh:
            n *= int(d)
        lname = key[:-len(".weight")] if key.endswith(".weight") else key[:-len(".bias")]
        a, _ = span.get(lname, (off, off))
        span[lname] = (a, off + n)
        off += n
    total = off
    target = total / float(n_buckets)
    buckets, cur, size = [], [], 0
    for name in layers:
        a, b = span[param_key(name, "weight")[:-len(".weight")]]
        cur.append(name)
        size += b - a
        if size >= target and len(buckets) < n_buckets - 1:
            buckets.append(cur)
            cur, size = [], 0
    if cur:
        buckets.append(cur)
    return sizes, span, buckets


class UNetGrad:
    """Two forward plans are supported: the fp32 plan with materialised upsample tensors (every activation is an fp32 plane),
    and the Q8 plan (mode f16f8, fused upsample) with fp32 twins: there the data gradients run on the fp16 + fp8 convolution
    kernel (dZ is written in the Q8 form by ssm_lrelu_bwd_q8), the weight gradients read the fp32 twins, and the pooled / upsampled
    operands the fused forward never materialised are recomputed in fp32 by `prepare`."""

    def __init__(self, plan):
        self.plan = plan
        self.hl8 = plan.hl8
        assert plan.hoist is None, "a hoisted inference plan (per-pair partial sums) cannot serve the backward"
        if self.hl8:
            assert plan.q8 and plan.twins, "the backward over an HL8 plan needs mode f16f8 with fp32 twins"
        else:
            assert not plan.fuse_up, "the fp32 plan must materialise the upsample tensors for the backward"
        self.B, self.dev = plan.B, plan.device
        self.g, self.dz, self.dzq, self.pk_t, self.grads, self.tw = {}, {}, {}, {}, {}, {}
        self.io = {n: (srcs, dst) for n, srcs, dst in _CONV_IO}
        # Parameter gradients leave the critical path: with `side` set (a HIP stream) every layer's weight/bias-gradient launches go
        # there, ordered after that layer's dZ by an event, while the data-gradient chain continues on the caller's stream.  Nothing
        # they read (dZ of the layer, forward activations) is rewritten before join(); their outputs are read after it.
        self.side, self._ev = None, {}
        self.more_sides, self._rr = [], 0          # further side streams the layers' weight gradients are dealt over ($SSM_WGRAD_STREAMS)
        # the f16f8 plan's weight gradients run on the split-bf16 matrix path; the exact plan's on fp32 MFMA
        self.split_wgrad = self.hl8 and os.environ.get("SSM_WGRAD", "bf16x3") != "f32"
        # every parameter gradient of the U-Net lives in ONE flat buffer (state-dict order), zeroed by one launch per backward; the
        # weight / bias gradient kernels accumulate into their slices (they add partial sums with atomics anyway)
        sizes, span, self.buckets = grad_layout(plan.layers, self.N_BUCKETS)
        self.flat = torch.empty(sum(int(torch.Size(sh).numel()) for _, sh in sizes), dtype=torch.float32, device=self.dev)
        off = 0
        for key, sh in sizes:
            n = int(torch.Size(sh).numel())
            self.grads[key] = self.flat[off:off + n].view(sh)
            off += n
        # Buckets of the flat buffer ("bucketed and overlapped with backward"; the partition itself: grad_layout).  When the
        # weight-gradient kernels of a bucket's last layer have been queued (side stream), the bucket is post-scaled (loss scale of the
        # f16f8 plan, 1/world) and handed to `sync.reduce` - the RCCL all-reduce of the decoder's gradients then runs while the
        # encoder's (and the other U-Net's) backward is still computing.
        self.sync, self.post_scale = None, 1.0
        self.layer_span = span
        self._pending = []
        # Winograd-domain weight gradients: per layer a [16, Cout, Cin] scratch of partial sums (one zeroed arena per U-Net; the finishing
        # launch of a bucket - G^T dU G added into the flat gradient buffer - leaves it zeroed for the next step)
        self.ww, self._ww_finish = {}, [None] * len(self.buckets)
        if WGRAD_WINO and getattr(plan, "wino", False) and not self.hl8:
            chosen = []
            for name, (ci, co, k) in plan.layers.items():
                s = layer_scale(name)
                srcs = self.io[name][0]
                cmin = min(plan.t[sn].C for sn in srcs) if all(sn in plan.t for sn in srcs) else ci
                # where it pays (profiles/r17d_wgradw_layers.txt, r17f_train_ab.txt): every workgroup adds 16 x its couts x cins partial sums
                # whatever its share of K, so the form wins on the layers with many tiles per (cout, cin) pair - the decoder's "a" layers
                # (Cin >= 2 Cout) and the 176 / 352-pixel maps - and ties or loses on the square 44 / 88-pixel layers, which keep the
                # direct kernel ($SSM_WGRAD_WINO=all: every supported layer)
                pays = WGRAD_WINO_ALL or ci >= 2 * co or plan.W // s >= 176
                if pays and hb.wgrad_wino_supported(cmin, co, plan.H // s, plan.W // s, k):
                    chosen.append((name, 16 * co * ci))
            arena = torch.zeros(sum(n for _, n in chosen), dtype=torch.float32, device=self.dev)
            off = 0
            for name, n in chosen:
                ci, co, _ = plan.layers[name]
                self.ww[name] = arena[off:off + n].view(16, co, ci)
                off += n
            for i, bucket in enumerate(self.buckets):
                ent = [(self.ww[n], self.grads[param_key(n, "weight")]) for n in bucket if n in self.ww]
                if ent:
                    self._ww_finish[i] = hb.WgradWinoFinish(ent, self.dev)

    N_BUCKETS = int(os.environ.get("SSM_GRAD_BUCKETS", "4"))

    def _span_of(self, name):
        """[a, b) of layer `name`'s weight + bias inside the flat buffer."""
        return self.layer_span[param_key(name, "weight")[:-len(".weight")]]

    def _arm(self):
        self._pending = [set(b) for b in self.buckets]

    def _layer_done(self, name):
        """Called when a layer's parameter-gradient kernels have been queued.  A completed bucket is finished on the FIRST side stream,
        behind every side stream that may carry one of its layers."""
        for i, pend in enumerate(self._pending):
            if name in pend:
                pend.discard(name)
                if not pend:
                    for sd in self._more():
                        hb.stream_wait(sd, self.side)
                    with torch.cuda.stream(self.side) if self.side is not None else contextlib.nullcontext():
                        if self._ww_finish[i] is not None:      # the bucket's Winograd-domain partial sums -> dW (behind them)
                            self._ww_finish[i].run()
                        a = self._span_of(self.buckets[i][0])[0]
                        b = self._span_of(self.buckets[i][-1])[1]
                        view = self.flat[a:b]
                        if self.post_scale != 1.0:
                            scale = self.post_scale
                            hb.host_op(lambda: view.mul_(scale))
                        if self.sync is not None:
                            sync = self.sync
                            hb.host_op(lambda: sync.reduce(view))
                return

    def join(self):
        """The caller's stream waits for the parameter gradients queued on the side stream."""
        if self.side is not None:
            hb.stream_wait(self.side, torch.cuda.current_stream())
            for sd in self._more():
                hb.stream_wait(sd, torch.cuda.current_stream())

    def _more(self):
        """The further side streams - none while a HIP graph is being captured (the opt-in Trainer(graphs=True) path: capture_end of
        ROCm 7.2 crashed with three forked streams, profiles/r22_backward_tests_3streams.txt; that path keeps the one side stream)."""
        if self.more_sides and not torch.cuda.is_current_stream_capturing():
            return self.more_sides
        return ()

    def act(self, name):
        """fp32 planes of a forward activation."""
        if not self.hl8:
            return self.plan.t[name]
        return self.plan.f32[name] if name in self.plan.f32 else self.tw[name]

    def prepare(self, cross=None):
        """Q8 plan: recompute in fp32 the pooled and the concatenated+upsampled conv inputs (weight-gradient operands)."""
        if not self.hl8:
            return
        lib, st, f32 = hb.load(), hb.stream_ptr(), self.plan.f32
        for conv, pname in _POOL_OF.items():
            src = f32[self.io[conv][1]]
            if pname not in self.tw:
                self.tw[pname] = hb.Planes(self.B, src.C, src.H // 2, src.W // 2, self.dev)
            hb.check(lib.ssm_avgpool2_fwd(src.view(), self.tw[pname].view(), self.B, src.C, src.H, src.W, st))
        for uname, (a, b) in _UP_SOURCES.items():
            A = f32[a]
            Bp = cross if (uname == "u7" and self.plan.cross) else (f32[b] if b else None)
            cb = Bp.C if Bp is not None else 0
            if uname not in self.tw:
                self.tw[uname] = hb.Planes(self.B, A.C + cb, 2 * A.H, 2 * A.W, self.dev)
            hb.check(lib.ssm_upsample2x_cat_fwd(A.view(), A.C, Bp.view() if Bp is not None else hb.NULL_VIEW, cb,
                                                self.tw[uname].view(), self.B, A.H, A.W, st))

    def _G(self, name, like=None, C=None):
        """Gradient buffer with the geometry of activation `like` (default: same name)."""
        if name not in self.g:
            ref = self.act(like or name) if (like or name) != "out" else self.plan.t["out"]
            self.g[name] = hb.Planes(self.B, C or ref.C, ref.H, ref.W, self.dev)
        return self.g[name]

    def refresh(self, state_dict, need_input_grad):
        """Repack the data-gradient filters (the weights change every optimizer step).  Q8 plan: all of them by one launch straight
        from the forward OIHW parameters (hb.PackBatch, transposed jobs); fp32 plan: layer by layer from the materialised
        transposed + flipped filters."""
        names = [n for n in self.plan.layers if not (n == "conv1a" and not need_input_grad)]
        if self.hl8:
            ws = [state_dict[param_key(n, "weight")] for n in names]
            if all(w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() for w in ws):
                scales = tuple(self.plan.scales[n] for n in names)
                key = (tuple(names), hb.PackBatch.key(ws), scales)
                if getattr(self, "_pack", None) is None or self._pack[0] != key:
                    if getattr(self, "_pack", None) is not None and self._pack[0][2] != scales:
                        self.pk_t = {}          # the forward plan re-chose its filter pre-scales: repack with them
                    entries = []
                    for n, w in zip(names, ws):
                        ci, co, k = self.plan.layers[n]
                        if n not in self.pk_t:
                            self.pk_t[n] = hb.PackedConv16(None, None, self.plan.W // layer_scale(n), q8=True, scale=self.plan.scales[n],
                                                           shape=(ci, co, k), device=self.dev)
                        entries.append((self.pk_t[n], w, None, True))
                    self._pack = (key, hb.PackBatch(entries, self.dev))
                self._pack[1].run()
                return
        ws = [state_dict[param_key(n, "weight")] for n in names]
        batch32 = (not self.hl8 and os.environ.get("SSM_PACK_BATCH", "1") != "0"
                   and all(w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() for w in ws))
        key32 = (tuple(names), hb.PackBatch.key(ws)) if batch32 else None
        if batch32 and getattr(self, "_pack32", None) is not None and self._pack32[0] == key32:
            self._pack32[1].run()          # every data-gradient filter straight from the forward parameters, one launch
            return
        self._pack32 = None
        for name in names:
            ci, co, k = self.plan.layers[name]
            s = layer_scale(name)
            w = state_dict[param_key(name, "weight")].to(device=self.dev, dtype=torch.float32)
            if self.hl8:
                self.pk_t[name] = hb.PackedConv16(transposed_filter(w), torch.zeros(ci, device=self.dev), self.plan.W // s, q8=True,
                                                  scale=self.plan.scales[name])
            else:
                # the data gradient of a 3x3 convolution is a 3x3 convolution with the transposed + flipped filter: on an f32w plan it
                # runs in the Winograd form like the forward (its "input channels" are the layer's output channels)
                use_w = getattr(self.plan, "wino", False) and hb.wino_supported(co, ci, self.plan.H // s, self.plan.W // s, k)
                cls = hb.PackedWino if use_w else hb.PackedConv
                if (use_w and (getattr(self.plan, "wino4", False) or DGRAD_WINO4) and hb.wino4_supported(co, ci, self.plan.H // s, self.plan.W // s, k)
                        and hb.wino4_preferred(co, ci, self.B, self.plan.H // s, self.plan.W // s, False)):
                    cls = hb.PackedWino4
                from .engine import WINO5, WINO7, WINO_SKIP, wino1d_enabled
                if getattr(self.plan, "wino1d", False) and wino1d_enabled(k) and hb.wino1d_supported(co, ci, self.plan.H // s, self.plan.W // s, k):
                    cls = hb.PackedWino1d          # (inference-style plans only: a training plan never sets wino1d)
                # r5: the data gradient of a 7x7 / 5x5 layer in the blocked two-dimensional forms of the inference plans (csrc/ssm_wino7.hip,
                # ssm_wino5.hip) where the transposed layer has whole 32-channel output blocks (conv1b, conv2a, conv2b).  A data gradient is
                # linear in dZ and the masks come from the forward: the gradient carries the form's own 3e-5 and the bars of
                # tests/test_hip_backward.py hold.  $SSM_TRAIN_DGRAD_BLOCKED=0: off; the plan-wide switches $SSM_WINO7=0 / $SSM_WINO5=0 /
                # $SSM_WINO_SKIP=<layer> restore the direct form here as they do in the forward
                skip = name in WINO_SKIP or "all" in WINO_SKIP
                if getattr(self.plan, "wino", False) and DGRAD_BLOCKED and not skip:
                    if WINO7 not in ("0", "") and (hb.wino7_supported(co, ci, self.plan.H // s, self.plan.W // s, k) or (k == 7 and 8 <= ci < 32)):
                        cls = hb.PackedWino7          # (stage 2's conv1a, 16 inputs: the transposed layer padded to one 32-channel block, see backward())
                    elif WINO5 not in ("0", "") and hb.wino5_supported(co, ci, self.plan.H // s, self.plan.W // s, k) and co % 4 == 0:
                        cls = hb.PackedWino5
                self.pk_t[name] = cls(transposed_filter(w), torch.zeros(ci, device=self.dev), self.B, self.plan.H // s, self.plan.W // s)
                if getattr(self.plan, "wino", False) and cls is hb.PackedConv:
                    self.pk_t[name].split_ok = True          # (mode f32w: see engine.UNetPlan.refresh_weights)
        if batch32:
            self._pack32 = (key32, hb.PackBatch32([(self.pk_t[n], w, None, True) for n, w in zip(names, ws)], self.dev))

    def _dzbuf(self, name):
        """dZ planes of a layer (channels padded to its data-gradient filter's chunk), built on first use."""
        if name not in self.dz:
            _, dst = self.io[name]
            co = self.plan.layers[name][1]
            Y = self.act(dst) if dst != "out" else self.plan.t["out"]
            pk = self.pk_t.get(name)
            self.dz[name] = hb.Planes(self.B, pk.cin_p if pk is not None else co, Y.H, Y.W, self.dev)
        return self.dz[name]

    def _layer(self, name, dy, dpool, dx, need_wgrad, act=True, dz_ready=False, fuse_next=None):
        """One convolution: dZ, parameter gradients, data gradient into `dx` (None: not needed).
        dz_ready: this layer's dZ was already written by the kernel that produced its upstream gradient (no ssm_lrelu_bwd launch).
        fuse_next: the layer whose OUTPUT is this layer's only input - its dZ is this layer's data gradient x LeakyReLU'(that output), and
        where the data-gradient kernel has the mask epilogue (F(2x2) / F(4x4), SSM_FLAG_MASK) it is written directly (`dx` stays
        untouched).  Returns True if it was: the caller passes dz_ready=True to that layer."""
        plan = self.plan
        srcs, dst = self.io[name]
        ci, co, k = plan.layers[name]
        Y = self.act(dst) if dst != "out" else plan.t["out"]
        pk = self.pk_t.get(name)
        cpad = pk.cin_p if pk is not None else co
        dzp = self._dzbuf(name)
        dz = dzp.slice(0, co)
        if dz_ready:
            pass          # (written by the data-gradient launch of the layer above / the upsample adjoint)
        elif self.hl8 and dx is not None:          # dZ also in the Q8 form: operand of the data-gradient convolution
            if name not in self.dzq:
                self.dzq[name] = hb.HPlanes(self.B, cpad, Y.H, Y.W, self.dev, q8=True)
            hb.check(hb.load().ssm_lrelu_bwd_q8(dy.view() if dy is not None else hb.NULL_VIEW,
                                                dpool.view() if dpool is not None else hb.NULL_VIEW, Y.view(), dz.view(),
                                                self.dzq[name].view(), self.B, co, Y.H, Y.W, 0.1, 1 if act else 0, hb.stream_ptr()))
        else:
            lrelu_bwd(dy, dpool, Y, dz, has_act=act)
        from .engine import UNetPlan
        tm = UNetPlan.timer
        flops = 2.0 * self.B * Y.H * Y.W * co * ci * k * k
        if need_wgrad:
            wk, bk = param_key(name, "weight"), param_key(name, "bias")
            sd = self.side
            if sd is not None and self._more():          # deal the layers over the side streams: their weight gradients are independent
                allsd = [self.side] + list(self.more_sides)
                sd = allsd[self._rr % len(allsd)]
                self._rr += 1
            if sd is not None:
                hb.stream_wait(torch.cuda.current_stream(), sd)      # dZ of this layer is complete on the caller's stream
            with torch.cuda.stream(sd) if sd is not None else contextlib.nullcontext():
                du = self.ww.get(name)
                if tm is not None:
                    e0, e1 = tm.span("wgrad", "s%d.%s" % (plan.stage, name), flops, issued=flops * (16.0 / 36.0 if du is not None else 1.0))
                    e0.record()
                off = 0
                for sname in srcs:
                    X = self.act(sname)
                    if du is not None:          # Winograd domain: partial sums into the layer's scratch, dW at the bucket's finishing launch
                        hb.wgrad_wino(X.view(), dz.view(), du, self.grads[bk] if off == 0 else None, self.B, X.C, co, Y.H, Y.W, ci, off)
                        off += X.C
                        continue
                    # fp32 path: the bias gradient rides in the first source's GEMM as one more column (no pass of its own over dZ)
                    wgrad(X, dz, self.grads[wk], k, ci_offset=off, zero_first=False, split=self.split_wgrad,
                          bias_acc=self.grads[bk] if off == 0 and not self.split_wgrad else None)
                    off += X.C
                assert off == ci, "%s: inputs cover %d of %d channels" % (name, off, ci)
                if self.split_wgrad:
                    bias_grad(dz, self.grads[bk], zero_first=False)
                if tm is not None:
                    e1.record()
            self._layer_done(name)
        if dx is not None:
            if tm is not None:
                from .engine import issued_factor
                e0, e1 = tm.span("dgrad", "s%d.%s" % (plan.stage, name), flops, issued=flops * (1.0 if self.hl8 else issued_factor(pk)))
                e0.record()
            fused = False
            if self.hl8:
                hb.conv2d_hl8(self.dzq[name].view(), cpad, None, 0, pk, None, dx.view(), None, self.B, Y.H, Y.W, lrelu=False)
            else:
                from .engine import conv_fn
                if fuse_next is not None and LRELU_FUSE and pk.algo in FUSE_ALGOS and plan.layers[fuse_next][1] == ci:
                    below = self.act(self.io[fuse_next][1])          # the output of the layer below = this layer's input
                    conv_fn(pk)(dzp.view(), cpad, None, 0, pk, self._dzbuf(fuse_next).slice(0, ci).view(), None, self.B, Y.H, Y.W, lrelu=False,
                                add=below.view(), mask=True)
                    fused = True
                else:
                    conv_fn(pk)(dzp.view(), cpad, None, 0, pk, dx.view(), None, self.B, Y.H, Y.W, lrelu=False)
            if tm is not None:
                e1.record()
            return fused
        return False

    def backward(self, d_out, need_wgrad=True, need_input_grad=False, cross_grad_out=None, c6_grad_init=None):
        """d_out: Planes with the gradient of final_conv's output (channels padded to the data-gradient chunk).
        cross_grad_out (stage 2): Planes that receives the gradient wrt the stage-1 bottleneck fed through the
        cross-skip.  c6_grad_init (stage 1): Planes already holding that gradient; the decoder's own gradient wrt
        conv6.1's output is added to it.  Returns the gradient of the input (Planes) or None."""
        L, G = self._layer, self._G
        plan = self.plan
        if need_wgrad:
            if self.side is not None:
                hb.stream_wait(torch.cuda.current_stream(), self.side)       # last step's consumers of the gradients are done
            with torch.cuda.stream(self.side) if self.side is not None else contextlib.nullcontext():
                hb.host_op(self.flat.zero_)
            for sd in self._more():          # (their first weight gradient must find the zeroed buffer)
                hb.stream_wait(self.side, sd)
            self._arm()
        fuse = LRELU_FUSE and not self.hl8

        def up(du, a_name, b_planes, conv_below, **kw):
            """Adjoint of cat + upsample; with the fusion on, the a-source's gradient leaves as dZ of `conv_below` (its only consumer).
            Returns dz_ready for that layer."""
            if fuse and not kw.get("acc_a"):
                upsample_cat_bwd(du, self._dzbuf(conv_below).slice(0, plan.layers[conv_below][1]), b_planes, mask_a=self.act(a_name), **kw)
                return True
            upsample_cat_bwd(du, G(a_name), b_planes, **kw)
            return False

        L("final_conv", d_out, None, G("tf"), need_wgrad, act=False)
        cat = G("cat_fuse", like="tf", C=plan.t["c11"].C + plan.t["c1"].C)
        L("fuse_conv", G("tf"), None, cat, need_wgrad)
        d_c11, d_c1 = cat.slice(0, plan.t["c11"].C), cat.slice(plan.t["c11"].C, plan.t["c1"].C)
        f = L("conv11b", d_c11, None, G("t11a"), need_wgrad, fuse_next="conv11a")
        L("conv11a", G("t11a"), None, G("u11"), need_wgrad, dz_ready=f)
        r = up(G("u11"), "c10", G("c2"), "conv10b")
        f = L("conv10b", G("c10"), None, G("t10a"), need_wgrad, dz_ready=r, fuse_next="conv10a")
        L("conv10a", G("t10a"), None, G("u10"), need_wgrad, dz_ready=f)
        r = up(G("u10"), "c9", G("c3"), "conv9b")
        f = L("conv9b", G("c9"), None, G("t9a"), need_wgrad, dz_ready=r, fuse_next="conv9a")
        L("conv9a", G("t9a"), None, G("u9"), need_wgrad, dz_ready=f)
        r = up(G("u9"), "c8", G("c4"), "conv8b")
        f = L("conv8b", G("c8"), None, G("t8a"), need_wgrad, dz_ready=r, fuse_next="conv8a")
        L("conv8a", G("t8a"), None, G("u8"), need_wgrad, dz_ready=f)
        r = up(G("u8"), "c7", G("c5"), "conv7b")
        f = L("conv7b", G("c7"), None, G("t7a"), need_wgrad, dz_ready=r, fuse_next="conv7a")
        L("conv7a", G("t7a"), None, G("u7"), need_wgrad, dz_ready=f)
        if c6_grad_init is not None:            # stage 1 with a cross-skip: add to the gradient stage 2 left there
            self.g["c6"] = c6_grad_init
            upsample_cat_bwd(G("u7"), self.g["c6"], None, acc_a=True)
            r = False
        elif plan.cross:                        # stage 2: second source of u7 = stage 1's bottleneck
            r = up(G("u7"), "c6", cross_grad_out, "conv6.1")
        else:
            r = up(G("u7"), "c6", None, "conv6.1")
        f = L("conv6.1", G("c6"), None, G("t6a"), need_wgrad, dz_ready=r, fuse_next="conv6.0")
        L("conv6.0", G("t6a"), None, G("p6"), need_wgrad, dz_ready=f)
        f = L("conv5b", G("c5"), G("p6"), G("t5a"), need_wgrad, fuse_next="conv5a")
        L("conv5a", G("t5a"), None, G("p5"), need_wgrad, dz_ready=f)
        f = L("conv4b", G("c4"), G("p5"), G("t4a"), need_wgrad, fuse_next="conv4a")
        L("conv4a", G("t4a"), None, G("p4"), need_wgrad, dz_ready=f)
        f = L("conv3b", G("c3"), G("p4"), G("t3a"), need_wgrad, fuse_next="conv3a")
        L("conv3a", G("t3a"), None, G("p3"), need_wgrad, dz_ready=f)
        f = L("conv2b", G("c2"), G("p3"), G("t2a"), need_wgrad, fuse_next="conv2a")
        L("conv2a", G("t2a"), None, G("p2"), need_wgrad, dz_ready=f)
        f = L("conv1b", d_c1, G("p2"), G("t1a"), need_wgrad, fuse_next="conv1a")
        # (a data gradient in the blocked 7x7 form writes whole 32-channel blocks: zeros beyond the layer's inputs)
        d_in = G("in", C=getattr(self.pk_t.get("conv1a"), "cout_p", None)) if need_input_grad else None
        L("conv1a", G("t1a"), None, d_in, need_wgrad, dz_ready=f)
        return d_in


class PairGrad:
    """Backward of one interpolation window on a PairEngine built with mode="f32", fuse_upsample=False, B1 == B2:
    loss gradient -> synthesis adjoint -> stage-2 U-Net -> compute_inputs adjoint -> stage-1 U-Net."""

    def __init__(self, engine):
        assert engine.B1 == engine.B2, "training runs one t per sample"
        assert not engine.hl8 or (engine.q8 and engine.twins), "an HL8 training engine must be mode f16f8 with fp32 twins"
        self.e = engine
        self.u1, self.u2 = UNetGrad(engine.s1), UNetGrad(engine.s2)
        if os.environ.get("SSM_WGRAD_STREAM", "1") != "0":
            self.u1.side = self.u2.side = torch.cuda.Stream(device=engine.device)
            extra = [torch.cuda.Stream(device=engine.device) for _ in range(max(0, int(os.environ.get("SSM_WGRAD_STREAMS", "2")) - 1))]
            self.u1.more_sides = self.u2.more_sides = extra
        B, H, W, dev = engine.B2, engine.H, engine.W, engine.device
        # gradient exchange (ssm_amd.dist.GradientAllReduce.attach): buckets are scaled by sync_scale (1/world) and reduced as they complete
        self.sync, self.sync_scale = None, 1.0
        self.dest = torch.empty(B, 4, H, W, dtype=torch.float32, device=dev)
        self.cr = torch.empty(B, dtype=torch.float32, device=dev)
        self.cw = torch.empty(B, dtype=torch.float32, device=dev)

    def owns(self, t):
        """True if tensor `t` is a slice of the flat gradient buffers the next backward rewrites."""
        for u in (self.u1, self.u2):
            a = u.flat.data_ptr()
            if a <= t.data_ptr() < a + 4 * u.flat.numel():
                return True
        return False

    def loss_scale(self, lambda_max, n_windows=1):
        """Power-of-two scale S applied to the loss gradient at the top of the backward and divided out of the parameter
        gradients at its end (exact in fp32, the chain is linear in dZ).  Only the f16f8 plan needs it: there dZ is carried as
        fp16 hi + e4m3(lo * 2^11), and the per-pixel loss gradient lambda / (3 H W) - 1.2e-5 for the shipped config at 224x224
        with batch 32 - is below fp16's smallest normal (6.1e-5): hi would be subnormal and the compensation part flush to zero.
        S brings the top-level coefficient to ~4 x the upstream gradient (4/B for a batch mean): the e4m3 compensation parts
        (lo * 2^11 ~ dZ/2) then sit in e4m3's normal range for the gradients that matter, while 448 (the clamp of the fp8(dZ)
        operand) and fp16's 65504 stay 2 and 4 orders of magnitude away."""
        if not self.e.hl8:
            return 1.0
        n = 3.0 * self.e.H * self.e.W * n_windows
        return 4.0 * 2.0 ** math.floor(math.log2(n / max(float(lambda_max), 1e-30)))

    def backward(self, sd1, sd2, target, g_losses, lambda_r, lambda_w, train_s1, train_s2, n_windows=1, dy_extra=None, lambda_p=0.0):
        """target [B,3,H,W]; g_losses [B,4] = upstream gradient of the [B,4] loss tensor (columns total, recon, warp,
        perceptual); returns {state-dict key: gradient} for the stages that train.  The warp-loss terms follow the
        FREEZE gating of scripts/models/losses.py:159-167.  dy_extra: ssm_view of d(perceptual term)/d(frame) ALREADY times
        loss_scale(max(lambda_r, lambda_w, lambda_p)), or None."""
        e = self.e
        lib = hb.load()
        st = hb.stream_ptr()
        B, H, W = e.B2, e.H, e.W
        n = 3.0 * H * W * n_windows
        S = self.loss_scale(max(lambda_r, lambda_w, lambda_p), n_windows)
        hb.host_op(lambda: (self.cr.copy_((g_losses[:, 0] + g_losses[:, 1]) * (S * lambda_r / n)),
                            self.cw.copy_((g_losses[:, 0] + g_losses[:, 2]) * (S * lambda_w / n))))
        target = target.contiguous()
        img6 = hb.view_of(e.img6)
        in16, out5, flow4 = e.s2.t["in"], e.s2.t["out"], e.s1.t["out"]
        need_s1 = train_s1
        for u in (self.u1, self.u2):          # every bucket of the flat gradient buffers is multiplied by this when it completes
            u.post_scale, u.sync = self.sync_scale / S, self.sync
        # (r6, measured and not kept: repacking the data-gradient filters on the side stream while the forward runs - 14.46 -> 16.2 ms per
        # step, profiles/r17n_train_ab2.txt: the side stream's launches at the head of the step delay the forward more than the 0.3 ms of
        # repack they take off the backward's chain)
        self.u2.refresh(sd2, need_input_grad=need_s1)
        self.u2.prepare(cross=e.s1.f32.get("c6") if e.hl8 else None)
        d_out5 = self.u2._G("out", C=self.u2.pk_t["final_conv"].cin_p)
        if e.hl8:       # the approximated flows live in their own fp32 tensor (the 16-channel input is Q8)
            ev = hb.view_of(e.est)
            est_view = ev
        else:
            est_view = in16.view(6)
        hb.check(lib.ssm_synthesize_bwd(img6, est_view, out5.view(), hb.view_of(target), e.t_dev.data_ptr(),
                                        self.cr.data_ptr(), self.cw.data_ptr(), dy_extra if dy_extra is not None else hb.NULL_VIEW,
                                        d_out5.view(), hb.view_of(self.dest), B, H, W,
                                        1 if train_s2 else 0, st))
        dcross = None
        if e.cross and need_s1:
            dcross = self.u1._G("c6")
        d_in16 = self.u2.backward(d_out5, need_wgrad=train_s2, need_input_grad=need_s1, cross_grad_out=dcross)
        grads = {}
        if train_s2:
            grads.update({"stage2." + k: v for k, v in self.u2.grads.items()})
        if need_s1:
            self.u1.refresh(sd1, need_input_grad=False)
            self.u1.prepare()
            d_flow4 = self.u1._G("out", C=self.u1.pk_t["final_conv"].cin_p)
            hb.check(lib.ssm_flowinterp_inputs_bwd(img6, flow4.view(), d_in16.view(), hb.view_of(self.dest), e.t_dev.data_ptr(),
                                                   self.cw.data_ptr(), d_flow4.view(), B, H, W, 1, st))
            self.u1.backward(d_flow4, need_wgrad=True, c6_grad_init=dcross)
            grads.update({"stage1." + k: v for k, v in self.u1.grads.items()})
        self.u2.join()
        return grads
