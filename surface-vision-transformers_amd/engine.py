"""Fused training step for the SiT hot path: the throughput counterpart of the reference's inner loops
(tools/train.py:280-291 and tools/pretrain.py:309-319):

    zero_grad -> forward -> loss -> backward -> [gradient all-reduce] -> optimizer.step

It drives the same C-ABI kernels as the drop-in modules, but without autograd: parameters, gradients
and optimizer state live in three flat fp32 buffers (the module's nn.Parameters are re-pointed at
views of them, so state_dict()/checkpoints keep working), every activation buffer is allocated
once.  Launch form (TrainEngine(use_graph=None) picks it): the 16-bit fused path (dim 192) is enqueued EAGERLY, ~110 launches
per step, because it forks 8 of 12 layers' weight gradients, the weight staging and the next step's gather onto a side stream
beside the backward chain (a forked step replayed from ONE hipGraph starts the chain's kernels late: 2.82 against 2.49 ms);
every other configuration is replayed from hipGraph(s).  With data parallelism every finished bucket of gradients is all-reduced
(RCCL) while the rest of backward runs -- from ONE stream the engine owns (`_ar_stream`): the collectives are issued as
synchronous operations with that stream current, and torch >= 2.8 launches a synchronous collective on the CURRENT stream (an
`async_op=True` one runs on the process group's internal stream behind an event: TrainEngine(dp_collective="group"), the control).

Unlike tools/train.py:293-296 nothing here synchronises with the host: `step()` returns a device
scalar (the loss) and never calls .item().
"""
import contextlib
import ctypes as C
import math

import torch

from . import ops
from . import runtime as rt
from .models.mpp import masked_patch_pretraining
from .models.sit import SiT

_ALIGN = 64  # floats (256 B): every parameter view is 16-byte aligned with room to spare


class FlatParams:
    """Flat fp32 storage for parameters / gradients / optimizer state of a module."""

    def __init__(self, module, device, grad_extra=0, order=None, share=None):
        """order: key(parameter) -> sortable; the buffer then holds the parameters sorted by it (stable: module order within
        one key).  The data-parallel engine orders by the point of the step at which a gradient becomes final, so that
        every all-reduce bucket is ONE contiguous range.  Nothing else depends on the order: the module's parameters are
        views (state_dict / checkpoints unchanged), the optimizer pass is elementwise.
        share: another FlatParams of the SAME module and order (SplitTrainEngine): this one uses ITS parameter buffer -- the module's
        parameters stay views of it, their .grad stays the first engine's -- and owns only a gradient buffer of the same layout."""
        self.params = []
        seen = set()
        for p in module.parameters():
            if id(p) not in seen:
                seen.add(id(p))
                self.params.append(p)
        if order is not None:
            self.params.sort(key=order)
        self.offsets, off = {}, 0
        for p in self.params:
            self.offsets[id(p)] = (off, p.numel())
            off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.total = off
        if share is not None:
            assert share.total == off and [id(p) for p in share.params] == [id(p) for p in self.params]
            self.flat = share.flat
            self.grad_all = torch.zeros(off + grad_extra, dtype=torch.float32, device=device)
            self.grad = self.grad_all[:off]
            self._extra_used = 0
            assert share.still_flat()
            return
        self.flat = torch.zeros(off, dtype=torch.float32, device=device)
        # grad_extra floats behind the gradients: per-step accumulators (loss, padded embedding gradient, ...) that are
        # zeroed together with the gradients by ONE fill (`grad_all.zero_()`); never part of an all-reduce range
        self.grad_all = torch.zeros(off + grad_extra, dtype=torch.float32, device=device)
        self.grad = self.grad_all[:off]
        self._extra_used = 0
        for p in self.params:
            o, n = self.offsets[id(p)]
            view = self.flat[o:o + n].view(p.shape)
            view.copy_(p.data.to(device=device, dtype=torch.float32))
            p.data = view
            p.grad = self.grad[o:o + n].view(p.shape)

    def g(self, p):
        o, n = self.offsets[id(p)]
        return self.grad[o:o + n].view(p.shape)

    def extra(self, shape):
        """A zero-initialised fp32 accumulator carved from the tail of the gradient buffer."""
        n = 1
        for d in shape:
            n *= d
        o = self.total + self._extra_used
        self._extra_used += (n + _ALIGN - 1) // _ALIGN * _ALIGN
        assert o + n <= self.grad_all.numel(), "FlatParams: grad_extra too small"
        return self.grad_all[o:o + n].view(shape)

    def offset(self, p):
        return self.offsets[id(p)][0]

    def still_flat(self):
        return all(p.data.data_ptr() == self.flat.data_ptr() + 4 * self.offsets[id(p)][0] for p in self.params)


def grad_write_stages(model, task, slices, head_deferred=False):
    """Backward stage after which each parameter's gradient is FINAL, keyed by id(parameter).

    Stage i < len(slices): backward slice i (slices run last layer first; segment 0 also holds forward, loss and the
    head's backward); stage len(slices): `_finish_backward`.  Derived from where the step WRITES each gradient, not
    from flat-buffer offsets:
      * layers lb..le-1 of slice i: weight / bias / LayerNorm gradients of those layers (the slice's one weight-gradient
        launch and its LayerNorm reduction close the slice);
      * mlp_head.*: the fused head + loss kernel of segment 0 (regression) -- or, with a side stream (head_deferred), the
        reduction of its per-sample terms in `_finish_backward`; untouched (zero) under MPP;
      * to_original.* (models/mpp.py:66-67,129): its weight gradient joins the weight-gradient launch of the slice that
        ends at layer 0 or runs in `_finish_backward` -> final only after finish, like the embedding, cls_token,
        pos_embedding and mask_token gradients."""
    sit = model.transformer if task == "mpp" else model
    n, finish = len(slices), len(slices)
    stage = {id(p): finish for p in model.parameters()}
    for i, (lb, le) in enumerate(slices):
        for layer in sit.transformer.layers[lb:le]:
            for p in layer.parameters():
                stage[id(p)] = i
    for p in sit.mlp_head.parameters():
        stage[id(p)] = finish if head_deferred else 0
    assert n >= 1
    return stage


def side_launch_groups(layer_begin, layer_end, side_layers, per_launch=2):
    """The layers whose weight + bias gradients each side launch of sitk_encoder_bwd_overlap carries (include/sitk.h, ABI 10):
    launch i = the `per_launch` layers below layer_end - per_launch i, the last one what is left of `side_layers`."""
    s = min(side_layers, layer_end - layer_begin)
    groups, top = [], layer_end
    while layer_end - top < s:
        n = min(per_launch, s - (layer_end - top))
        groups.append(list(range(top - n, top)))
        top -= n
    return groups


def side_bucket_sizes(n_launches, per_bucket):
    """Side launches per early all-reduce bucket as a list: an int k -> buckets of k consecutive launches (the last one
    smaller); a sequence -> those sizes, and side launches it does not cover travel with the FINAL bucket (behind the finish
    stage, which joins the side stream)."""
    if isinstance(per_bucket, int):
        k = max(1, per_bucket)
        return [min(k, n_launches - i) for i in range(0, n_launches, k)]
    sizes = [int(v) for v in per_bucket]
    if any(v < 1 for v in sizes) or sum(sizes) > n_launches:
        raise ValueError(f"dp_bucket_launches {sizes}: sizes >= 1 that cover at most the {n_launches} side launches")
    return sizes


def grad_write_stages_side(model, task, groups, per_bucket=1):
    """The side-stream form's counterpart of grad_write_stages: stage b < n_early = behind the LAST side launch of early bucket b
    (the Linear weights and biases of its launches' layers -- a side launch writes nothing else); stage n_early = behind the
    finish stage: every LayerNorm parameter (one reduction at the end of backward sums their partials), the layers whose weight
    gradients run in the tail launch behind the chain, side launches no early bucket covers, the patch embedding, cls_token,
    pos_embedding, mlp_head.* and, under MPP, to_original.* and mask_token.  per_bucket: see side_bucket_sizes."""
    sit = model.transformer if task == "mpp" else model
    sizes = side_bucket_sizes(len(groups), per_bucket)
    final = len(sizes)
    bucket_of = {}
    i = 0
    for b, n in enumerate(sizes):
        for _ in range(n):
            bucket_of[i] = b
            i += 1
    stage = {id(p): final for p in model.parameters()}
    for i, layers in enumerate(groups):
        for l in layers:
            for name, p in sit.transformer.layers[l].named_parameters():
                if ".norm." not in "." + name:
                    stage[id(p)] = bucket_of.get(i, final)
    return stage


PROBE_RATIO = 1.12        # chain with the candidate parked / chain with it idle: above this the candidate is rejected


def probe_stream(stream, main=None, reps=2):
    """sitk_stream_probe (include/sitk.h) of a torch stream against `main` (default: the current stream): the best of `reps` runs as
    {'free_us', 'blocked_us', 'done_us', 'release_us', 'ok'}.  ok = blocked behind an event the stream does not slow a chain of
    dependent launches on `main` (<= PROBE_RATIO x the same chain with the stream idle: measured 1.03 - 1.06 x for a harmless stream,
    1.22 - 1.28 x when `main` is the lowest-priority side stream and the two queues share a dispatch pipe, 1.6 - 2.6 x when `main` is
    the null stream and they do) AND its work ran beside that chain (finished within 150 us of its release; a stream that shares
    the chain's hardware QUEUE finishes behind the chain's end)."""
    main = torch.cuda.current_stream(stream.device) if main is None else main
    best = None
    for _ in range(reps):
        v = [C.c_float() for _ in range(4)]
        rt.check(rt.lib.sitk_stream_probe(main.cuda_stream, stream.cuda_stream, *[C.byref(x) for x in v]))
        r = dict(free_us=v[0].value, blocked_us=v[1].value, done_us=v[2].value, release_us=v[3].value)
        if best is None or r["blocked_us"] / r["free_us"] < best["blocked_us"] / best["free_us"]:
            best = r
    best["ok"] = bool(best["blocked_us"] <= PROBE_RATIO * best["free_us"] and best["done_us"] <= best["release_us"] + 150.0)
    return best


def pick_bucket_stream(device, priority=0, candidates=8, victims=()):
    """The stream the early all-reduce buckets are issued from, chosen by MEASUREMENT: candidates are taken from torch's stream pool
    one after the other (they land on the process's hardware queues in turn) and probed against the current stream; the first one
    that is harmless while blocked -- for the current stream and for every stream in `victims` -- and concurrent while running
    wins.  Returns (stream, [probe results]); if none passes, the least harmful one (the results say so: 'ok' False)."""
    tried = []
    for _ in range(candidates):
        st = torch.cuda.Stream(device=device, priority=priority)
        if any(st.cuda_stream == t[0].cuda_stream for t in tried):
            break                                         # the pool has wrapped around
        r = probe_stream(st)
        r["worst_ratio"] = r["blocked_us"] / r["free_us"] + (0.0 if r["done_us"] <= r["release_us"] + 150.0 else 10.0)
        # the candidate must also leave the OTHER streams of the step alone: parked, it delays the dispatches of every queue that
        # shares its dispatch pipe -- the side stream's few launches (weight gradients, reductions, column sums: ~20 dispatches
        # x ~35 us) were the victim in every slow data-parallel run of rounds 4 - 6
        for i, v in enumerate(victims):
            rv = probe_stream(st, main=v)
            r[f"victim{i}_free_us"], r[f"victim{i}_blocked_us"] = rv["free_us"], rv["blocked_us"]
            r["ok"] = bool(r["ok"] and rv["blocked_us"] <= PROBE_RATIO * rv["free_us"])
            r["worst_ratio"] = max(r["worst_ratio"], rv["blocked_us"] / rv["free_us"])
        tried.append((st, r))
        if r["ok"]:
            break
    st, _ = min(tried, key=lambda t: (not t[1]["ok"], t[1]["worst_ratio"]))      # (none passed: the least harmful one)
    return st, [dict(r, stream=f"{t.cuda_stream:#x}", chosen=(t is st)) for t, r in tried]


def grad_bucket_plan(fp, stage, n_slices, limit=None):
    """All-reduce ranges of the flat gradient buffer per point of the step: plan[i] (i < n_slices - 1) is issued right
    after backward slice i, plan[n_slices - 1] after `_finish_backward` (the last slice's gradients travel with the
    finish stage: nothing is left to overlap them with).  Adjacent parameters of one point merge into one range (the
    alignment padding between them rides along), so every float of the buffer is reduced exactly once and never before
    the kernel that writes it has been enqueued.  limit: the floats from `limit` on belong to parameters the optimizer does not
    touch (TrainEngine(optimize="sit")): they are not reduced."""
    last = n_slices - 1
    plan = [[] for _ in range(n_slices)]
    order = sorted(fp.params, key=lambda p: fp.offsets[id(p)][0])
    limit = fp.total if limit is None else limit
    for k, p in enumerate(order):
        lo = fp.offsets[id(p)][0]
        hi = fp.offsets[id(order[k + 1])][0] if k + 1 < len(order) else fp.total
        if lo >= limit:
            continue
        pt = min(stage[id(p)], last)
        if plan[pt] and plan[pt][-1][1] == lo:
            plan[pt][-1] = (plan[pt][-1][0], hi)
        else:
            plan[pt].append((lo, hi))
    return plan


class TrainEngine:
    """One fused train step of a SiT (task='regression') or of masked patch pre-training (task='mpp').

    input_layout: 'surface' -> step(x) takes raw channels-last surfaces (B, 40962, C) and gathers the
                  patches on the GPU; 'patched' -> the reference's (B, C, P, V) layout.
    normalise:    (mean, std) per channel -> (x - mean) / std fused into the gather (tools/preprocessing.py:72);
                  surface layout only.
    keep_grads:   True -> the parameters' .grad still hold this step's gradients after step() (the buffers are then
                  zeroed by a fill at the start of the next step); False (default) -> the optimizer pass zeroes every
                  gradient it consumes (optimizer.zero_grad() of tools/train.py:288 folded into optimizer.step()).

    wgrad_overlap: number of layers whose weight gradients run on a side stream BESIDE the rest of the backward chain, on
                  the CUs its one-wave kernels leave idle (sitk_encoder_bwd_overlap); 0 = off.  Needs eager launches.
    head_deferred: with a side stream, the sum of the head's per-sample gradient terms and of the loss runs THERE beside the
                  chain's tail (sitk_head_loss_fwd_bwd_deferred + sitk_head_finalize); False = behind the head kernel on the main stream.
    prefetch_gather: with a side stream, enqueue the patch gather of a regression step there (it then runs beside the previous
                  step's tail); False = in front of the patch embedding on the main stream.  Inputs must then reach the
                  engine through load_batch() / step(x, ...) / step(indices=...), which order that gather behind their copy
                  (a direct write into `eng.inp` on another stream is not seen by the side stream).  The gather hides behind
                  the previous step's tail only when the batch is static or selected by HOST indices of a resident data set
                  (bench.py's headline; -11 us per step): after load_batch() / step(x, ...) / device indices it has to wait
                  for that copy, which sits behind the whole previous step on the main stream -- no overlap, one more event
                  (bench.py reports this form too: also.new_batch_every_step).
    optimize:     task='mpp' only.  'all' (default): the optimizer updates every parameter of the pre-training module that receives a
                  gradient -- the encoder, the patch embedding, cls_token / pos_embedding AND the head of models/mpp.py:66,74
                  (to_original.*, mask_token), i.e. torch.optim.X(ssl.parameters()).  'sit': the scope of the REFERENCE loop,
                  tools/pretrain.py:267-280, which builds its optimizer over model.parameters() -- the SiT alone: to_original.* and
                  mask_token keep their initial values for the whole run.  Their gradients are still produced (the
                  weight-gradient launch and the column sums carry them) but neither applied nor all-reduced.  In BOTH modes
                  mlp_head.* is left alone under MPP: its gradient is None in the reference (the head is not on the MPP path), and
                  torch's optimizers skip such parameters -- momentum, weight decay and all.
    use_graph:    True = the step is replayed from hipGraph(s) (one per segment); False = eager launches.
                  None (default) with wgrad_overlap None: the faster of the two forms measured for the configuration --
                  eager + 8 of 12 layers on the side stream for the 16-bit fused path on one GPU (dim 192); under a process
                  group the same path runs its `dp_side` form: the SAME launch sequence plus all-reduce buckets behind the
                  side launches (see dp_bucket_launches); hipGraph replay without a side stream everywhere else (other widths,
                  f32, use_graph=True).
    dp_bucket_launches: data-parallel form of the fused path: side launches per early all-reduce bucket -- an int, or a list of
                  bucket sizes (side launches a list does not cover travel with the final bucket); None = [all but the last,
                  the last].  dp_channels: RCCL channels (= workgroups of an all-reduce) the weight-gradient launch behind
                  the chain leaves CUs for (16; set NCCL_MAX_NCHANNELS to the same value before creating the process group).
                  dp_stream_priority: priority of the stream the early buckets are reduced from (0; a high-priority stream
                  beside the chain cost a whole step in every measurement: profiles/r05_dp_budget.txt).
    dp_collective: where a bucket's all-reduce runs.  'stream' (default): on the engine's bucket stream `_ar_stream`, EVERY bucket
                  of the step (the final one behind an event of the main stream's finish stage) -- issued with async_op=False under
                  that stream, which torch >= 2.8 launches on the current stream (profiles/r06_dp_streams.txt: rocprofv3 trace);
                  the main stream waits for ONE event of the bucket stream in front of the optimizer pass.  The stream is picked
                  by measurement (pick_bucket_stream).  'group': async_op=True, i.e. on the process group's internal stream behind
                  an event of the issuing stream (rounds 2 - 5; kept as the control of tools/dp_cu_budget.py).
    wgrad_overlap_group: layers per side launch (2; 1 - 3).

    The learning rate (and Adam's step count) live in device memory: `set_lr()` takes effect in captured graphs too,
    so the schedulers of tools/pretrain.py:42-50 can drive the engine.  `load_dataset()` keeps a whole data set
    resident in HBM; `step(indices=...)` then assembles the batch on the GPU (tools/train.py:97-113,282).
    """

    def __init__(self, model, batch_size, *, task="regression", input_layout="surface", loss="mse", optimizer="sgd",
                 lr=1e-5, momentum=0.9, weight_decay=0.0, nesterov=False, betas=(0.9, 0.999), eps=1e-8,
                 process_group=None, bwd_slices=None, use_graph=None, device=None, normalise=None, keep_grads=False,
                 wgrad_overlap=None, prefetch_gather=True, wgrad_overlap_cus=None, head_deferred=True, dp_channels=None, optimize=None,
                 wgrad_overlap_group=None, dp_stream_priority=0, dp_bucket_launches=None, dp_collective="stream", dp_final_on="bucket", _share=None, _defer_optimizer=False):
        if task == "mpp":
            assert isinstance(model, masked_patch_pretraining)
            self.ssl, self.sit = model, model.transformer
        else:
            assert isinstance(model, SiT)
            self.ssl, self.sit = None, model
        sit = self.sit
        self.task, self.layout, self.loss_kind = task, input_layout, loss
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.module = model.to(self.device)
        self.B, self.P, self.V, self.Cc = batch_size, sit.num_patches, sit.num_vertices, sit.num_channels
        self.N, self.D, self.K = self.P + 1, sit.dim, sit.patch_dim
        self.dtype = rt.dtype_code(sit.compute_dtype)
        self.tdt = rt.torch_dtype(self.dtype)
        # f16 compute mode (IEEE half: 5 exponent bits): backward runs on a loss-scaled gradient stream.  {S, 1 / S} live in
        # device memory; regression: the fused head + loss launch picks S = 2^k from the batch's largest |d loss / d logits|
        # every step; MPP: a constant from the loss denominator (set below).  The optimizer divides S out again.
        self.loss_scaled = self.dtype == rt.F16
        self.gscale = torch.ones((2,), dtype=torch.float32, device=self.device)
        self.ld = ops.pad64(self.K)
        self.ncls = sit.mlp_head[1].weight.shape[0]
        self.pool_mean = int(sit.pool == "mean")
        tr = sit.transformer
        if tr.p_dropout > 0 or sit.dropout.p > 0:
            raise rt.SitkError("TrainEngine: dropout > 0 is not implemented on the fused path")
        if tr.depth and (tr.layers[0][0].fn.dim_head != 64 or not tr.layers[0][0].fn.project_out):
            raise rt.SitkError("TrainEngine: the fused path needs dim_head = 64 and a projected attention output "
                               "(the modules run other head widths stage by stage)")
        self.depth = tr.depth
        self.opt = dict(kind=optimizer, lr=lr, momentum=momentum, wd=weight_decay, nesterov=nesterov, betas=betas, eps=eps)
        self.keep_grads = keep_grads
        # {lr, beta1^t, beta2^t, t} in device memory: read by the optimizer kernels, so captured graphs follow set_lr()
        self.hyper = torch.tensor([lr, 1.0, 1.0, 0.0], dtype=torch.float64, device=self.device)
        # gradient elements the optimizer pass found not finite and skipped (device counter; `nonfinite_count` reads it).  The
        # guard is on in the loss-scaled f16 mode only: bf16 / f32 behave like the reference's optimizer.step() (tools/train.py:291)
        self.nonfinite = torch.zeros((1,), dtype=torch.int32, device=self.device)
        self.norm = None
        if normalise is not None:
            if input_layout != "surface":
                raise rt.SitkError("TrainEngine: normalise= needs input_layout='surface' (it is fused into the gather)")
            mean, std = (torch.as_tensor(t, dtype=torch.float32).reshape(-1).to(self.device) for t in normalise)
            assert mean.numel() == self.Cc and std.numel() == self.Cc
            self.norm = (mean.contiguous(), std.contiguous())
        self.dataset = None                              # (x_all, targets_all) resident in HBM, see load_dataset()
        self.idx = torch.zeros((batch_size,), dtype=torch.int32, device=self.device)
        self._replay_randoms = False
        self.pg = process_group
        self.world = 1 if process_group is None else torch.distributed.get_world_size(process_group)
        # the data-parallel form of the step (backward slices, bucketed all-reduce between captured segments): whenever a
        # process group is given -- also a ONE-rank group, which runs every collective of that form on the real backend
        # (the only way to put RCCL under the engine on a one-GPU box: tests/test_dp_gpu.py)
        self.dp = process_group is not None
        if dp_collective not in ("stream", "group"):
            raise rt.SitkError("TrainEngine: dp_collective is 'stream' or 'group'")
        self.dp_collective = dp_collective
        self.dp_final_on = dp_final_on          # "bucket": the final bucket from the bucket stream too; "main": from the main stream (A/B)
        if self.dp and dp_collective == "stream" and torch.distributed.get_backend(process_group) == "nccl":
            # (c10d before 2.8 launched EVERY NCCL collective on the process group's internal stream: the bucket stream would only
            # carry the events; the arrangement the stand-in study measured is the one torch >= 2.8 produces)
            ver = tuple(int(v) for v in torch.__version__.split("+")[0].split(".")[:2])
            if ver < (2, 8):
                raise rt.SitkError(f"TrainEngine(dp_collective='stream') needs torch >= 2.8 (synchronous NCCL collectives on the "
                                   f"current stream); this is {torch.__version__}: pass dp_collective='group'")
        self._unscale_in_place = keep_grads or self.dp
        self.nsteps = 0

        # backward slices (last layer first) and the gradient ranges that become final after each
        tr = sit.transformer
        fused = self.dtype != rt.F32 and bool(rt.lib.sitk_mlp_fused_supported(self.D, tr.mlp_dim, self.dtype))
        # Data parallelism on the 16-bit fused path, eager launches (the default there): the SAME launch sequence as on one GPU
        # -- one backward call, the first 2 / 3 of the layers' weight gradients on the side stream, two layers per side launch,
        # the rest in one tail launch behind the chain -- plus all-reduce buckets made of side launches: the library records an
        # event behind each launch (sitk_overlap_wait_side_launch), and the flat buffers are ordered by write stage so that a
        # bucket's gradients are one contiguous range (round 4: two slices, the first bucket -- 58 % of the bytes -- final only
        # 40 us after the chain; now a side launch's 3.5 MB are final from ~1.4 ms into a 2.4 ms step on, the last bucket is the
        # tail's; which launches share a bucket: dp_bucket_launches below).
        self.dp_side = bool(self.dp and fused and use_graph is not True and bwd_slices is None and tr.depth >= 2
                            and (wgrad_overlap is None or wgrad_overlap > 0))
        self.wgrad_overlap_group = int(wgrad_overlap_group) if wgrad_overlap_group else 2
        n_side_layers = min(tr.depth - 1, max(1, round(2 / 3 * tr.depth))) if wgrad_overlap is None else min(tr.depth - 1, int(wgrad_overlap))
        self._side_groups = side_launch_groups(0, tr.depth, n_side_layers, self.wgrad_overlap_group) if self.dp_side else []
        if self.dp_side:
            wgrad_overlap = None
        if optimize not in (None, "all", "sit") or (optimize == "sit" and task != "mpp"):
            raise rt.SitkError("TrainEngine: optimize is 'all' or 'sit' (task='mpp' only)")
        self.optimize = (optimize or "all") if task == "mpp" else "all"
        # parameters the optimizer pass does not touch live at the END of the flat buffers (see `optimize`)
        frozen = set()
        if task == "mpp":
            frozen |= {id(p) for p in sit.mlp_head.parameters()}
            if self.optimize == "sit":
                frozen |= {id(p) for p in self.ssl.to_original.parameters()} | {id(self.ssl.mask_token)}
        # Side launches per early all-reduce bucket.  Measured with a stand-in of RCCL's footprint and SURVEY section 5's wire time
        # on the bucket's own stream (profiles/r05_dp_budget.txt): every collective costs the step ~30 us whatever its size or
        # duration (a 16-workgroup kernel that wants whole CUs finds them only at a kernel boundary of the chain), while an early
        # bucket's wire time hides -- one launch per bucket (4 + 1 collectives) +280 - 300 us per step, two per bucket +177 - 186,
        # ALL side launches in one early bucket (beside the tail launch) +137 - 143, [all but the last, the last] +144 - 166 -- and
        # with the wire at HALF the assumed rate +290 against +236 - 251: the default is the latter, one collective more for a
        # first bucket that is on the wire a whole side launch before the chain ends.
        # (an int = launches per bucket; a list = the early buckets' sizes, uncovered launches travel with the final bucket: [3] =
        # the first three side launches early -- on the wire a whole side launch before the chain ends --, the fourth with the tail)
        n_launches = len(self._side_groups)
        self.dp_bucket_launches = dp_bucket_launches if dp_bucket_launches else ([n_launches - 1, 1] if n_launches >= 2 else 1)
        self._bucket_sizes = side_bucket_sizes(len(self._side_groups), self.dp_bucket_launches) if self.dp_side else []
        self._n_early = len(self._bucket_sizes)
        side_stage = grad_write_stages_side(self.module, task, self._side_groups, self.dp_bucket_launches) if self.dp_side else {}
        # _share (SplitTrainEngine): another engine of the same module whose parameter buffer this one uses; this one then runs
        # forward + backward into a gradient buffer of its own and leaves the optimizer pass to its owner
        self._defer_optimizer = bool(_defer_optimizer) or _share is not None
        self.fp = FlatParams(self.module, self.device, grad_extra=self.D * self.ld + 4 * _ALIGN + self.D,
                             order=lambda p: (id(p) in frozen, side_stage.get(id(p), 0)),
                             share=(_share.fp if isinstance(_share, TrainEngine) else None))
        self.n_opt = min([self.fp.offsets[i][0] for i in frozen], default=self.fp.total)
        dev, f32 = self.device, torch.float32
        B, P, N, D, K, ld = self.B, self.P, self.N, self.D, self.K, self.ld
        self.cfg = ops.encoder_cfg(B, N, D, tr.depth, tr.heads, tr.mlp_dim, self.dtype)
        self.acts, self.scratch = ops.encoder_workspace(self.cfg, dev)
        layers = tr.layer_tensors()
        self.Pa = ops.layer_param_array([[p.data for p in layer] for layer in layers])
        self.Ga = ops.layer_param_array([[self.fp.g(p) for p in layer] for layer in layers])
        self.inp = torch.zeros((B, 40962, self.Cc) if input_layout == "surface" else (B, self.Cc, P, self.V), dtype=f32, device=dev)
        self.table = sit.patch_table(dev) if input_layout == "surface" else None
        self.tokens = torch.zeros((B * P, ld), dtype=self.tdt, device=dev)
        self.w_embed = torch.zeros((D, ld), dtype=self.tdt, device=dev)
        self.x0 = torch.empty((B * N, D), dtype=f32, device=dev)
        self.xL = torch.empty((B * N, D), dtype=f32, device=dev)
        self.dx = torch.empty((B * N, D), dtype=f32, device=dev)
        self.dW_embed = self.fp.extra((D, ld))          # per-step accumulators live behind the gradients: one fill zeroes all
        self.dx_c = torch.empty((B * N, D), dtype=self.tdt, device=dev)     # compute-dtype d(x_0) for the embedding's dW
        self._embed_wgrad_done = False
        self.loss_acc = self.fp.extra((1,))             # accumulated by the loss kernels; zeroed with the gradients
        self.loss = torch.zeros((1,), dtype=f32, device=dev) if not keep_grads else self.loss_acc
        self._loss_extra_idx = (self.loss_acc.data_ptr() - self.fp.grad_all.data_ptr()) // 4 - self.fp.total
        if task == "regression":
            self.target = torch.zeros((B, self.ncls), dtype=f32, device=dev)
            self.logits = torch.empty((B, self.ncls), dtype=f32, device=dev)
            self.dlogits = torch.empty((B, self.ncls), dtype=f32, device=dev)
            self.head_ws = torch.empty(rt.lib.sitk_head_ws_floats(B, D, self.ncls), dtype=f32, device=dev)
        else:
            ssl = self.ssl
            self.n_mask = math.ceil(ssl.mask_prob * P)
            # d loss / d batch_out = 2 (out - target) / (B n_mask K): times S = 2^floor(log2(B n_mask K / 2)) it is (out - target)
            # times a factor in (2, 4], whatever the batch size
            self.mpp_scale = float(2 ** math.floor(math.log2(B * self.n_mask * K / 2))) if self.loss_scaled else 1.0
            self.gscale.copy_(torch.tensor([self.mpp_scale, 1.0 / self.mpp_scale]))
            self.tok32 = torch.zeros((B * P, K), dtype=f32, device=dev)
            self.enc_out = torch.empty((B * P, D), dtype=self.tdt, device=dev)
            self.wo_c = torch.zeros((ld, D), dtype=self.tdt, device=dev)     # to_original (K, D), zero rows up to ld: the
            self.bo_pad = torch.zeros((ld,), dtype=f32, device=dev)          # GEMM runs on N = ld columns (16-byte stores)
            self.wo_t = torch.empty((D, ops.pad8(K)), dtype=self.tdt, device=dev)
            self.we_t = torch.empty((K, D), dtype=torch.float32 if self.loss_scaled else self.tdt, device=dev)   # embedding weight^T (for d mask_token)
            self.out_pad = torch.empty((B * P, ld), dtype=f32, device=dev)   # batch_out in rows of ld floats
            self.out = self.out_pad[:, :K]                                   # (B * P, K) view: models/mpp.py:129's batch_out
            self.dout_c = torch.zeros((B * P, ld), dtype=self.tdt, device=dev)   # d batch_out, compute dtype; pad columns stay 0
            self._extra_wgrad_done = False
            self.masked = torch.zeros((B * P,), dtype=torch.uint8, device=dev)
            self.repl = torch.zeros((B * P,), dtype=torch.uint8, device=dev)
            self.swap = torch.zeros((B * P,), dtype=torch.uint8, device=dev) if ssl.swap_prob > 0 else None
            self.rpatch = torch.zeros((B * P,), dtype=torch.int32, device=dev) if ssl.swap_prob > 0 else None
            self.replaced_full = torch.zeros((B, N), dtype=torch.uint8, device=dev)
            self.rsum = self.fp.extra((1, D))
            # Philox stream of the on-device draws: {seed, draws so far}; every rank its own seed
            rank = torch.distributed.get_rank(process_group) if process_group is not None else 0
            self.rng_state = torch.tensor([(torch.initial_seed() + 0x9E3779B97F4A7C15 * rank) & 0x7FFFFFFFFFFFFFFF, 0],
                                          dtype=torch.int64, device=dev)
        if _share is not None:
            self.state = [None, None]                   # (the owner of the shared parameters holds the optimizer state)
        elif optimizer == "sgd":
            self.state = [torch.zeros_like(self.fp.flat)] if momentum != 0 else [None]
        elif optimizer in ("adam", "adamw"):
            self.state = [torch.zeros_like(self.fp.flat), torch.zeros_like(self.fp.flat)]
        else:
            raise ValueError(optimizer)

        if self.dp_side:
            self.slices = [(0, tr.depth)]
            bwd_slices = 1
        if bwd_slices is None:
            # every extra slice costs ~35 us (its own weight-gradient launch and reduction: 2.77 / 2.82 / 2.86 / 2.89 ms per
            # step for 1 / 2 / 3 / 4 slices, SiT-tiny B = 64) and hides that fraction of the gradient all-reduce less
            # wide models, measured on a one-rank RCCL group (tools/dp_wide_probe.py, profiles/r06_dp_wide.txt): SiT-base 39.22 / 39.23 /
            # 39.27 ms for 1 / 2 / 3 slices (three: the exposed final bucket is the smallest), SiT-small 15.59 / 15.51 / 16.28 (its
            # 4-layer weight-gradient launches fill the chip's rounds badly: two slices)
            bwd_slices = 1 if not self.dp else min(2 if self.D == 384 else 3, tr.depth)
        if not self.dp_side:
            bounds = [round(i * tr.depth / bwd_slices) for i in range(bwd_slices + 1)]
            self.slices = [(bounds[i], bounds[i + 1]) for i in range(bwd_slices)][::-1]
        auto = wgrad_overlap is None
        if auto:
            # measured on MI355X (tiny, B = 64, eager launches, 256 x 192 weight-gradient tiles): 2.56 ms without, 2.45 with 7 of
            # 12 layers on the side stream, 2.43 with 8, 2.46 with 9, 2.52 with 10 (the side stream then finishes after the chain's
            # own tail launch); with round 2's 128 x 192 tiles the optimum was 7
            wgrad_overlap = round(2 / 3 * tr.depth) if (use_graph is not True and fused and not self.dp and bwd_slices == 1) else 0
        if wgrad_overlap > 0 and (self.dp or bwd_slices != 1 or use_graph):
            # A step with a forked side stream is enqueued eagerly: replayed from a hipGraph, ROCm 7.2 runs the two branches on
            # two queues but the chain's own kernels then start late (2.82 ms per step against 2.49 eager; the host needs ~0.4
            # ms to enqueue a step of 2.5 ms, so eager launches cost nothing: 2.54 against 2.55 ms without the side stream)
            raise rt.SitkError("wgrad_overlap needs one GPU, one backward slice and eager launches (use_graph=False)")
        if self.dp_side:
            wgrad_overlap = sum(len(g) for g in self._side_groups)
        max_side = int(wgrad_overlap)
        self.wgrad_overlap = int(wgrad_overlap)
        # workgroups of one side launch (two layers): 42 = the CUs the dim-192 chain's one-wave kernels leave idle
        self.wgrad_overlap_cus = int(wgrad_overlap_cus) if wgrad_overlap_cus else 42
        self._overlap, self.side_stream_probe = (self._create_overlap(max_side) if wgrad_overlap > 0 else (None, []))
        if self._overlap and self.wgrad_overlap_group != 2:
            rt.check(rt.lib.sitk_overlap_set_group(self._overlap, self.wgrad_overlap_group))
        self._head_deferred = bool(self._overlap) and bool(head_deferred) and task == "regression"
        self._side = rt.lib.sitk_overlap_stream(self._overlap) if self._overlap else None
        self._side_torch = torch.cuda.ExternalStream(self._side, device=self.device) if self._overlap else None
        # With a side stream the patch gather of a regression step -- which reads the input batch and the patch table, no
        # parameter -- is enqueued THERE, in front of the fork that orders the weight staging behind the previous step's
        # optimizer pass: it runs as soon as the side stream has finished the previous step's work, i.e. beside that step's
        # tail launch and optimizer, instead of in front of the patch embedding (30 us of the chain).  The previous step's
        # tail still reads ITS tokens (the patch embedding's weight gradient): two token buffers, alternating.
        self._prefetch = bool(prefetch_gather and self._overlap and self.layout == "surface" and task == "regression")
        if self._prefetch:
            self._tok_bufs = (self.tokens, torch.zeros_like(self.tokens))
            self._ev_gather, self._ev_inp, self._inp_dirty = torch.cuda.Event(), torch.cuda.Event(), False
        if self.dp_side:
            # Bucket i is final behind side launch i (an event the library records on the side stream); its all-reduce is issued
            # from a small stream of its own that waits for THAT event only -- not from the side stream's context, whose tail by
            # then holds later launches.  The one weight-gradient launch behind the chain leaves the channels' CUs free.
            self.dp_channels = int(dp_channels) if dp_channels else 16
            rt.check(rt.lib.sitk_overlap_set_tail_cus(self._overlap, max(64, 256 - self.dp_channels)))
        if self.dp:
            # the ONE stream every early bucket is reduced from (see dp_collective), and the event behind its last collective.
            # WHICH stream matters (round 6, profiles/r06_dp_streams.txt): the bucket stream spends most of the step blocked behind a
            # side launch's event, and a blocked stream whose hardware queue shares a dispatch pipe with the main stream's delays every
            # dispatch of the chain (~35 us each; +0.7 .. +2.3 ms per step in rounds 4 - 5), one that shares its QUEUE runs behind the
            # chain instead of beside it.  Placement follows the process's stream creation order, so it is measured, not assumed.
            with torch.cuda.device(self.device):
                self._ar_stream, self.dp_stream_probe = pick_bucket_stream(
                    self.device, int(dp_stream_priority), victims=([self._side_torch] if self._side_torch is not None else []))
            self._ar_done = torch.cuda.Event()
            self._ar_used = False
        if use_graph is None:
            use_graph = not self._overlap
        if self.dp_side:
            self.bucket_plan = grad_bucket_plan(self.fp, side_stage, self._n_early + 1, limit=self.n_opt)
        else:
            self.bucket_plan = grad_bucket_plan(self.fp, grad_write_stages(self.module, task, self.slices,
                                                                           head_deferred=self._head_deferred), len(self.slices),
                                                limit=self.n_opt)
        self.use_graph = use_graph
        self._graphs = None
        self._pending = []

    def _create_overlap(self, max_side, attempts=4):
        """The overlap object (side stream + events) -- with the side stream's PLACEMENT measured: the side stream spends much of the
        step blocked behind fork events of the chain, and a blocked stream on a hardware queue that shares a dispatch pipe with the
        main stream's delays every dispatch of the chain (see pick_bucket_stream; rounds 3 - 5 met it as "engines created later in
        one process measure up to 2 ms slower" and "two side streams: the chain 1 230 instead of 890 us").  A side stream that fails
        the probe is kept alive while the next one is created (so that one lands on another hardware queue) and destroyed afterwards."""
        tried = []
        with torch.cuda.device(self.device):
            for _ in range(attempts):
                ov = rt.lib.sitk_overlap_create(max_side, self.wgrad_overlap_cus, 1)
                if not ov:
                    raise rt.SitkError(f"sitk_overlap_create: {rt.lib.sitk_last_error().decode()}")
                side = torch.cuda.ExternalStream(rt.lib.sitk_overlap_stream(ov), device=self.device)
                r = probe_stream(side)
                tried.append((ov, r))
                if r["ok"]:
                    break
        best = min(tried, key=lambda t: (not t[1]["ok"], t[1]["blocked_us"] / t[1]["free_us"]))
        for ov, _ in tried:
            if ov is not best[0]:
                rt.lib.sitk_overlap_destroy(ov)
        return best[0], [dict(r, chosen=(ov is best[0])) for ov, r in tried]

    def __del__(self):
        ov, self._overlap = getattr(self, "_overlap", None), None
        if ov and rt is not None and getattr(rt, "lib", None) is not None:      # (module globals may be gone at interpreter exit)
            rt.lib.sitk_overlap_destroy(ov)

    # ---------------------------------------------------------------------------------------------
    def _s(self):
        return rt.stream_ptr()

    def _stage_beside(self):
        """With a side stream: the compute-dtype weight copies of all layers (one launch, ~16 us) are staged THERE, beside the
        gather and the patch embedding on the main stream; _encoder_forward joins.  Returns the `save` flags of the forward."""
        if not self._overlap:
            return 1
        rt.check(rt.lib.sitk_overlap_fork(self._overlap, self._s()))     # behind the previous step's optimizer pass
        rt.check(rt.lib.sitk_encoder_stage_weights(C.byref(self.cfg), self.Pa, self.acts.data_ptr(), self.acts.numel(), self._side))
        return 3

    def _encoder_forward(self, save):
        if self._overlap:
            rt.check(rt.lib.sitk_overlap_join(self._overlap, self._s()))
        rt.check(rt.lib.sitk_encoder_fwd(C.byref(self.cfg), self.Pa, self.x0.data_ptr(), self.xL.data_ptr(), self.acts.data_ptr(),
                                         self.acts.numel(), self.scratch.data_ptr(), self.scratch.numel(), save, self._s()))

    def _forward_regression(self):
        sit, L, s = self.sit, rt.lib, self._s()
        B, P, N, D, K, ld, dt = self.B, self.P, self.N, self.D, self.K, self.ld, self.dtype
        lin = sit.to_patch_embedding[1]
        if self.keep_grads:
            self.fp.grad_all.zero_()                    # gradients + loss + padded embedding gradient
        if self._prefetch:
            self.tokens = self._tok_bufs[self.nsteps & 1]
            if self._inp_dirty:                         # load_batch() / the index copy of this step, enqueued on the main stream
                self._side_torch.wait_event(self._ev_inp)
                self._inp_dirty = False
            self._gather(self.tokens, ld, dt, stream=self._side)
            self._ev_gather.record(self._side_torch)
        save = self._stage_beside()
        if self._prefetch:
            torch.cuda.current_stream(self.device).wait_event(self._ev_gather)
        elif self.layout == "surface":
            self._gather(self.tokens, ld, dt)
        else:
            rt.check(L.sitk_patchify(self.inp.data_ptr(), self.tokens.data_ptr(), B, self.Cc, P, self.V, ld, dt, s))
        self._embed_forward(self.tokens)
        self._encoder_forward(save)
        ln, fc = sit.mlp_head[0], sit.mlp_head[1]
        g = self.fp.g
        if self._head_deferred:
            # With a side stream the sum of the head's per-sample gradient terms (and of the loss) leaves the chain: backward
            # waits for dx only; _embed_backward runs sitk_head_finalize on the side stream beside the chain's tail.
            rt.check(L.sitk_head_loss_fwd_bwd_deferred(self.xL.data_ptr(), ln.weight.data_ptr(), ln.bias.data_ptr(),
                                                       fc.weight.data_ptr(), fc.bias.data_ptr(), self.target.data_ptr(),
                                                       self.logits.data_ptr(), self.dx.data_ptr(), B, N, D, self.ncls,
                                                       self.pool_mean, int(self.loss_kind == "l1"), self.head_ws.data_ptr(),
                                                       self.gscale.data_ptr() if self.loss_scaled else None, s))
            return
        # pool + head + loss + their backward: one launch (dx = d(loss)/d(x_L) for every row comes out of it)
        rt.check(L.sitk_head_loss_fwd_bwd(self.xL.data_ptr(), ln.weight.data_ptr(), ln.bias.data_ptr(), fc.weight.data_ptr(),
                                          fc.bias.data_ptr(), self.target.data_ptr(), self.logits.data_ptr(),
                                          self.loss_acc.data_ptr(), self.dx.data_ptr(), g(ln.weight).data_ptr(),
                                          g(ln.bias).data_ptr(), g(fc.weight).data_ptr(), g(fc.bias).data_ptr(), B, N, D,
                                          self.ncls, self.pool_mean, int(self.loss_kind == "l1"), self.head_ws.data_ptr(),
                                          self.gscale.data_ptr() if self.loss_scaled else None, s))

    def _gather(self, out, ld, dt, stream=None):
        """patch gather of the batch: from the static input buffer, or -- after load_dataset() -- straight from the resident
        data set through the step's sample indices (their labels ride along); per-channel normalisation fused if set."""
        L, s = rt.lib, (stream if stream is not None else self._s())
        mean, std = (self.norm[0].data_ptr(), self.norm[1].data_ptr()) if self.norm else (None, None)
        if self.dataset is not None:
            x_all, t_all = self.dataset
            tgt = self.task == "regression" and t_all is not None
            rt.check(L.sitk_gather_tokens_idx(x_all.data_ptr(), self.idx.data_ptr(), self.table.data_ptr(), mean, std,
                                              out.data_ptr(), t_all.data_ptr() if tgt else None,
                                              self.target.data_ptr() if tgt else None, self.ncls if tgt else 0, self.B, 40962,
                                              self.Cc, self.P, self.V, ld, dt, s))
        elif self.norm:
            rt.check(L.sitk_gather_tokens_norm(self.inp.data_ptr(), self.table.data_ptr(), mean, std, out.data_ptr(), self.B,
                                               40962, self.Cc, self.P, self.V, ld, dt, s))
        else:
            rt.check(L.sitk_gather_tokens(self.inp.data_ptr(), self.table.data_ptr(), out.data_ptr(), self.B, 40962, self.Cc,
                                          self.P, self.V, ld, dt, s))

    def _embed_forward(self, tokens):
        sit, L, s = self.sit, rt.lib, self._s()
        B, P, N, D, K, ld, dt = self.B, self.P, self.N, self.D, self.K, self.ld, self.dtype
        lin = sit.to_patch_embedding[1]
        rt.check(L.sitk_stage_weight(lin.weight.data_ptr(), D, K, self.w_embed.data_ptr(), ld, None, 0, dt, s))
        pos = sit.pos_embedding.data.view(-1, D)
        ops.gemm_nt(tokens, self.w_embed, self.x0, dt, M=B * P, N=D, K=ld, epilogue=ops.EPI_BIAS_RES, bias=lin.bias.data,
                    aux=pos, omap=(P, N, 1), auxmap=(P, 0, 1))
        # the cls rows of x0 depend on no product: with a side stream (forked in _stage_beside, joined in _encoder_forward) they
        # are written there, beside the patch embedding
        rt.check(L.sitk_embed_cls_rows(self.x0.data_ptr(), sit.cls_token.data_ptr(), pos.data_ptr(), B, N, D,
                                       self._side if self._overlap else s))

    def _embed_backward(self, tokens):
        sit = self.sit
        B, P, N, D, K, ld, dt = self.B, self.P, self.N, self.D, self.K, self.ld, self.dtype
        lin = sit.to_patch_embedding[1]
        g = self.fp.g
        if not self._embed_wgrad_done:      # (shapes outside the batched large-tile path: padded scratch, then a strided copy)
            ops.gemm_wgrad(self.dx, tokens, self.dW_embed, dt, db=g(lin.bias), M=B * P, N=D, K=ld, dymap=(P, N, 1))
            g(lin.weight).copy_(self.dW_embed[:, :K])
        gpos = g(sit.pos_embedding).view(-1)[:N * D]
        # d pos_embedding / d cls_token: column sums of the chain's final dx.  With a side stream (which
        # sitk_encoder_bwd_overlap left behind the chain's last kernel) they run beside the tail weight-gradient launch.
        rt.check(rt.lib.sitk_colsum_f32_dup(self.dx.data_ptr(), B, N * D, N * D, gpos.data_ptr(),
                                            g(sit.cls_token).data_ptr(), D, self._side if self._overlap else self._s()))
        if self._head_deferred and self.task == "regression":      # (see _forward_regression: the head's gradients and the loss)
            ln, fc = sit.mlp_head[0], sit.mlp_head[1]
            rt.check(rt.lib.sitk_head_finalize(self.head_ws.data_ptr(), B, D, self.ncls, g(ln.weight).data_ptr(),
                                               g(ln.bias).data_ptr(), g(fc.weight).data_ptr(), g(fc.bias).data_ptr(),
                                               self.loss_acc.data_ptr(), self._side))
        if self.task == "mpp":
            # d mask_token = (sum of dx over the replaced rows) W_embed: reads the chain's final dx like the sums above, so with a
            # side stream it runs there too, beside the tail weight-gradient launch (27 us at the end of the step otherwise)
            with (torch.cuda.stream(self._side_torch) if self._overlap else contextlib.nullcontext()):
                ops.masked_colsum(self.dx, self.replaced_full.view(-1), None, self.rsum, "f32")
                # f16 mode: rsum is a sum over ALL replaced rows (~25 k at config 5) of a gradient that carries the loss scale
                # -- far beyond f16's range as a GEMM operand: this 1 x D x K product runs in the exact-f32 kernel there
                pdt = rt.F32 if self.loss_scaled else dt
                rt.check(rt.lib.sitk_stage_weight(lin.weight.data_ptr(), D, K, None, 0, self.we_t.data_ptr(), D, pdt, self._s()))
                ops.gemm_nt(self.rsum, self.we_t, g(self.ssl.mask_token).view(1, K), pdt, M=1, N=K, K=D)
        if self._overlap:
            rt.check(rt.lib.sitk_overlap_join(self._overlap, self._s()))      # every gradient of the step is behind this point

    def _forward_mpp(self):
        ssl, sit, L, s = self.ssl, self.sit, rt.lib, self._s()
        B, P, N, D, K, ld, dt = self.B, self.P, self.N, self.D, self.K, self.ld, self.dtype
        if self.keep_grads:
            self.fp.grad_all.zero_()                    # gradients + loss + padded embedding gradient + rsum
        save = self._stage_beside()
        # the four random tensors of models/mpp.py:25-43,95-110 are drawn on the device (Philox; same distribution, not the
        # reference's stream: the parity path replays the reference's generator order through sitk.models.mpp instead)
        p_swap = ssl.swap_prob / (1 - ssl.replace_prob) if ssl.swap_prob > 0 else 0.0
        if not self._replay_randoms:                    # (set_randoms(): the mask buffers already hold the tensors to replay)
            rt.check(L.sitk_mpp_draw(self.rng_state.data_ptr(), self.masked.data_ptr(), rt.ptr(self.swap), rt.ptr(self.rpatch),
                                     self.repl.data_ptr(), self.replaced_full.data_ptr(), B, P, self.n_mask, p_swap,
                                     ssl.replace_prob, s))
        mt = ssl.mask_token.data_ptr()
        if self.layout == "surface":
            # gather + corruption in one pass: clean fp32 tokens (the regression target) and corrupted compute-dtype tokens
            mean, std = (self.norm[0].data_ptr(), self.norm[1].data_ptr()) if self.norm else (None, None)
            src = self.dataset[0] if self.dataset is not None else self.inp
            rt.check(L.sitk_mpp_gather_corrupt(src.data_ptr(), self.table.data_ptr(),
                                               self.idx.data_ptr() if self.dataset is not None else None, mean, std,
                                               self.masked.data_ptr(), rt.ptr(self.swap), rt.ptr(self.rpatch), self.repl.data_ptr(),
                                               mt, self.tok32.data_ptr(), self.tokens.data_ptr(), self.rng_state.data_ptr(), B,
                                               40962, self.Cc, P, self.V, ld, dt, s))
        else:
            rt.check(L.sitk_patchify(self.inp.data_ptr(), self.tok32.data_ptr(), B, self.Cc, P, self.V, K, rt.F32, s))
            rt.check(L.sitk_mpp_corrupt(self.tok32.data_ptr(), self.masked.data_ptr(), rt.ptr(self.swap), rt.ptr(self.rpatch),
                                        self.repl.data_ptr(), mt, self.tokens.data_ptr(), B, P, K, ld, dt, s))
            self.rng_state[1:2] += 1
        self._embed_forward(self.tokens)
        self._encoder_forward(save)
        # to_original on tokens 1..P (models/mpp.py:129) and masked MSE (models/mpp.py:132)
        lo = ssl.to_original
        rt.check(L.sitk_stage_weight(lo.weight.data_ptr(), K, D, self.wo_c.data_ptr(), D, self.wo_t.data_ptr(),
                                     self.wo_t.shape[1], dt, s))
        # encoder output rows 1..P in the compute dtype: operand of to_original and, later, of its weight gradient (which
        # joins the encoder's one weight-gradient launch: sitk_encoder_bwd_extra, see _backward_slice / _finish_backward)
        rt.check(L.sitk_cast_rows(self.xL.data_ptr() + 4 * D, N * D, self.enc_out.data_ptr(), P * D, B, P * D, dt, s))
        rt.check(L.sitk_cast_rows(lo.bias.data_ptr(), K, self.bo_pad.data_ptr(), ld, 1, K, rt.F32, s))
        # batch_out on ld (zero-weight padded) columns: the weight-resident streaming GEMM with whole-row stores; then one
        # row-layout pass for the masked squared error and its gradient in the compute dtype
        ops.gemm_nt(self.enc_out, self.wo_c, self.out_pad, dt, M=B * P, N=ld, K=D, bias=self.bo_pad)
        rt.check(L.sitk_mpp_loss_fwd_bwd_ld(self.out_pad.data_ptr(), ld, self.tok32.data_ptr(), K, self.masked.data_ptr(),
                                            self.loss_acc.data_ptr(), self.dout_c.data_ptr(), ld, dt, B * P, K, B * self.n_mask,
                                            self.mpp_scale, s))
        self.dx.zero_()
        ops.gemm_nt(self.dout_c, self.wo_t, self.dx, dt, M=B * P, N=D, K=self.wo_t.shape[1], omap=(P, N, 1))

    def _backward_slice(self, lb, le):
        if lb == 0 and self.dx_c is not None:
            # the slice that ends at layer 0 also carries the patch embedding's weight gradient (one launch for all)
            lin = self.sit.to_patch_embedding[1]
            extra = None
            if self.task == "mpp":
                lo = self.ssl.to_original
                extra = [ops.wgrad_desc(self.dout_c, self.enc_out, self.fp.g(lo.weight), db=self.fp.g(lo.bias), M=self.B * self.P,
                                        N=self.K, K=self.D)]
            self._embed_wgrad_done, self._extra_wgrad_done = ops.encoder_bwd_embed(
                self.cfg, self.Pa, self.Ga, self.x0, self.dx, self.acts, self.scratch, lb, le, self.tokens, self.fp.g(lin.weight),
                self.fp.g(lin.bias), self.dx_c, self.P, extra=extra,   # written straight into the (D, K) gradient (tokens: zero pad to ld)
                overlap=self._overlap)
            return
        ops.encoder_bwd(self.cfg, self.Pa, self.Ga, self.x0, self.dx, self.acts, self.scratch, layer_begin=lb, layer_end=le)

    def _finish_backward(self):
        self._embed_backward(self.tokens)
        if self.task == "mpp":
            ssl, sit, L, s = self.ssl, self.sit, rt.lib, self._s()
            D, K, dt = self.D, self.K, self.dtype
            lin = sit.to_patch_embedding[1]
            if not self._extra_wgrad_done:
                lo = ssl.to_original
                ops.gemm_wgrad(self.dout_c, self.enc_out, self.fp.g(lo.weight), dt, db=self.fp.g(lo.bias), M=self.B * self.P,
                               N=K, K=D)

    @property
    def last_randoms(self):
        """The random tensors of the most recent MPP step (models/mpp.py's names), e.g. to replay it through the modules."""
        B, P = self.B, self.P
        out = {"corrupted_sequence": self.masked.view(B, P).bool().clone(), "replace_draw": self.repl.view(B, P).bool().clone()}
        if self.swap is not None:
            out.update(swap_draw=self.swap.view(B, P).bool().clone(), random_patches=self.rpatch.view(B, P).long().clone())
        return out

    def set_randoms(self, randoms):
        """Replay the given random tensors (models/mpp.py:85-110's names, as `last_randoms` returns them: corrupted_sequence,
        swap_draw, random_patches, replace_draw; each (B, P)) in every following step instead of drawing on the device --
        e.g. masks captured from the reference, or another rank's draws.  None: draw on the device again.  The loss keeps
        the reference's fixed denominator, so every row must select exactly ceil(mask_prob P) patches."""
        if self.task != "mpp":
            raise rt.SitkError("set_randoms: task='mpp' only")
        if bool(randoms) != self._replay_randoms:
            self._graphs = None                          # the draw node comes / goes: capture again
        self._replay_randoms = bool(randoms)
        if not randoms:
            return
        B, P, dev = self.B, self.P, self.device
        m = torch.as_tensor(randoms["corrupted_sequence"]).to(dev).reshape(B, P).bool()
        if not bool((m.sum(1) == self.n_mask).all()):
            raise rt.SitkError(f"set_randoms: every row of corrupted_sequence must select {self.n_mask} patches")
        r = torch.as_tensor(randoms["replace_draw"]).to(dev).reshape(B, P).bool()
        self.masked.copy_(m.reshape(-1).to(torch.uint8))
        self.repl.copy_(r.reshape(-1).to(torch.uint8))
        if self.swap is not None:
            rp = torch.as_tensor(randoms["random_patches"]).to(dev).reshape(-1)
            if int(rp.min()) < 0 or int(rp.max()) >= P:
                raise rt.SitkError(f"set_randoms: random_patches must lie in [0, {P})")
            self.swap.copy_(torch.as_tensor(randoms["swap_draw"]).to(dev).reshape(-1).to(torch.uint8))
            self.rpatch.copy_(rp.to(torch.int32))
        self.replaced_full.zero_()
        self.replaced_full[:, 1:] = (m & r).to(torch.uint8)

    @property
    def nonfinite_count(self):
        """Gradient ELEMENTS skipped so far because they were not finite (a host read: syncs).  f16 mode only (always 0 in bf16 /
        f32, which run unguarded like the reference): not zero means an intermediate overflowed behind the loss scale; the
        parameters are intact (the optimizer pass skips such elements; the loss scale itself does not react)."""
        return int(self.nonfinite.item())

    def set_lr(self, lr):
        """New learning rate from the next step on (a device-side write: captured graphs read it from memory)."""
        self.opt["lr"] = float(lr)
        self.hyper[0:1].fill_(float(lr))

    def _optimizer(self, other=None):
        """The fused optimizer pass over this engine's gradient buffer -- or, with `other` (SplitTrainEngine: the engine of the other
        half batch, same parameters, its own gradient buffer), over the MEAN of the two buffers: each half's loss is a mean over its
        own samples, so the whole batch's gradient is (g_a + g_b) / 2 and its loss (l_a + l_b) / 2; the pass clears both."""
        o, fp, L, s = self.opt, self.fp, rt.lib, self._s()
        scale = (0.5 if other is not None else 1.0) / self.world
        # the loss scale S of the f16 mode: divided out by the optimizer pass itself (1 / S read from device memory), unless
        # the gradients were already unscaled in place (kept gradients; data parallelism, where every rank has its own S)
        inv_s = self.gscale[1:2].data_ptr() if self.loss_scaled and not self._unscale_in_place else None
        zero = int(not self.keep_grads)
        # the pass covers the first n_opt floats; the gradients of the parameters it leaves alone (behind them, see `optimize`)
        # are cleared with the per-step accumulators, never applied
        n = self.n_opt
        n_extra = fp.grad_all.numel() - n if zero else 0
        keep_idx = self._loss_extra_idx + (fp.total - n)
        keep_dst = self.loss.data_ptr() if zero else None
        guard = self.nonfinite.data_ptr() if self.loss_scaled else None
        g2 = inv_s2 = None
        if other is not None:
            assert other.fp.grad_all.numel() == fp.grad_all.numel() and other.fp.flat.data_ptr() == fp.flat.data_ptr()
            g2 = other.fp.grad_all.data_ptr()
            inv_s2 = other.gscale[1:2].data_ptr() if other.loss_scaled and not other._unscale_in_place else None
        keep_scale = 0.5 if other is not None else 1.0
        if o["kind"] == "sgd":
            rt.check(L.sitk_sgd_step_dev(fp.flat.data_ptr(), fp.grad_all.data_ptr(), rt.ptr(self.state[0]), n,
                                         self.hyper.data_ptr(), o["momentum"], o["wd"], int(o["nesterov"]), scale, zero,
                                         n_extra, keep_idx, keep_dst, inv_s, guard, g2, inv_s2, keep_scale, s))
        else:
            rt.check(L.sitk_adam_step_dev(fp.flat.data_ptr(), fp.grad_all.data_ptr(), self.state[0].data_ptr(),
                                          self.state[1].data_ptr(), n, self.hyper.data_ptr(), o["betas"][0],
                                          o["betas"][1], o["eps"], o["wd"], int(o["kind"] == "adamw"), scale, zero, n_extra,
                                          keep_idx, keep_dst, inv_s, guard, g2, inv_s2, keep_scale, s))

    # ---- segments: [fwd + loss + head/backward slice 0], [slice 1], ..., [finish + optimizer] ------------
    def _segment_fns(self):
        fwd = self._forward_mpp if self.task == "mpp" else self._forward_regression
        segs = []
        for i, (lb, le) in enumerate(self.slices):
            if i == 0:
                segs.append(lambda lb=lb, le=le: (fwd(), self._backward_slice(lb, le)))
            else:
                segs.append(lambda lb=lb, le=le: self._backward_slice(lb, le))
        return segs

    def _allreduce(self, lo, hi):
        """Sum-all-reduce of grad[lo:hi] on the CURRENT stream's behalf: dp_collective 'stream' -> the collective's kernel is
        launched on the current stream itself (a synchronous c10d operation: nothing to wait for afterwards but the stream);
        'group' -> on the process group's internal stream behind an event, the work handle kept until the optimizer pass."""
        if not self.dp:
            return
        if self.loss_scaled:                # every rank scaled by its own S: reduce unscaled gradients
            self.fp.grad[lo:hi].mul_(self.gscale[1])
        if self.dp_collective == "stream":
            torch.distributed.all_reduce(self.fp.grad[lo:hi], group=self.pg, async_op=False)
        else:
            self._pending.append(torch.distributed.all_reduce(self.fp.grad[lo:hi], group=self.pg, async_op=True))

    def _allreduce_early(self, ranges, behind_side_launch=None):
        """An early bucket: reduced from the bucket stream, behind side launch `behind_side_launch` (an event the library recorded
        on the side stream) or, without one, behind everything enqueued on the main stream so far."""
        if not ranges:
            return
        if behind_side_launch is None and self.dp_collective == "group":
            for lo, hi in ranges:               # (the process group's stream waits for the issuing one by itself)
                self._allreduce(lo, hi)
            return
        if behind_side_launch is None:
            self._ar_stream.wait_stream(torch.cuda.current_stream(self.device))
        else:
            rt.check(rt.lib.sitk_overlap_wait_side_launch(self._overlap, behind_side_launch, self._ar_stream.cuda_stream))
        with torch.cuda.stream(self._ar_stream):
            for lo, hi in ranges:
                self._allreduce(lo, hi)
        self._ar_used = True

    def _join_collectives(self):
        """The main stream behind every collective of the step (in front of the optimizer pass)."""
        if self._ar_used:
            self._ar_done.record(self._ar_stream)
            torch.cuda.current_stream(self.device).wait_event(self._ar_done)
            self._ar_used = False
        for w in self._pending:
            w.wait()
        self._pending.clear()

    def _run(self, fn, idx):
        if not self.use_graph:
            fn()
            return
        if self._graphs is None:
            self._graphs = {}
        if idx not in self._graphs:
            fn()                                   # eager warm-up (also validates arguments)
            # nothing else may touch the device while a stream captures (capture_error_mode "global"): a collective of an
            # earlier segment still copying on another thread / stream makes the capture fail now and then (gloo does).
            for w in self._pending:
                w.wait()
            torch.cuda.synchronize()               # (also drains the bucket stream: a synchronous gloo collective has returned by now)
            gr = torch.cuda.CUDAGraph()
            # thread_local: calls made by OTHER threads (the process group's watchdog polling its events) must not
            # invalidate this thread's capture
            with torch.cuda.graph(gr, capture_error_mode="thread_local"):
                fn()
            self._graphs[idx] = gr
            return                                 # the eager run above already did this step's work
        self._graphs[idx].replay()

    def load_batch(self, x, target=None):
        self.inp.copy_(x, non_blocking=True)
        if target is not None:
            self.target.copy_(target.reshape(self.target.shape), non_blocking=True)
        self._mark_input()

    def _mark_input(self):
        """The step's gather runs on the side stream (see _prefetch): it must see the input / index copy enqueued just now."""
        if getattr(self, "_prefetch", False):
            self._ev_inp.record(torch.cuda.current_stream(self.device))
            self._inp_dirty = True

    def load_dataset(self, x_all, targets_all=None):
        """Keep a whole data set resident in HBM: x_all (S, 40962, C) fp32 raw (un-normalised if `normalise` was given)
        surfaces, targets_all (S, n_classes) labels.  From then on step(indices=...) selects the batch by index inside
        the gather kernel; per step only the B int32 indices cross PCIe (tools/train.py:97-113,282-283 move the data)."""
        if self.layout != "surface":
            raise rt.SitkError("load_dataset: needs input_layout='surface'")
        x_all = torch.as_tensor(x_all, dtype=torch.float32).to(self.device).contiguous()
        if x_all.dim() != 3 or x_all.shape[1] != 40962 or x_all.shape[2] != self.Cc:
            raise rt.SitkError(f"load_dataset: expected (S, 40962, {self.Cc}), got {tuple(x_all.shape)}")
        t_all = None
        if targets_all is not None:
            t_all = torch.as_tensor(targets_all, dtype=torch.float32).to(self.device).reshape(x_all.shape[0], -1).contiguous()
            assert self.task != "regression" or t_all.shape[1] == self.ncls
        self.dataset = (x_all, t_all)
        self.idx.zero_()
        self._mark_input()
        self._graphs = None                               # the gather node changes: capture again

    def unload_dataset(self):
        """Back to the static input buffer (load_batch / step(x, target))."""
        self.dataset = None
        self._graphs = None

    def step(self, x=None, target=None, indices=None):
        """Runs one optimisation step on the batch in the static input buffers (or on x/target if given; or on the
        samples `indices` (B,) of the resident data set).  Returns the device tensor holding the loss of this step
        (no host sync)."""
        if indices is not None:
            if self.dataset is None:
                raise rt.SitkError("step(indices=...) needs load_dataset() first")
            if x is not None or target is not None:
                raise rt.SitkError("step: pass either indices= (resident data set) or x/target, not both")
            idx = torch.as_tensor(indices).reshape(-1)
            S = self.dataset[0].shape[0]
            if idx.numel() != self.B:
                raise rt.SitkError(f"step(indices=...): expected {self.B} sample indices, got {idx.numel()}")
            # the gather kernels index the resident data set with these values unchecked: validate them here (host values:
            # on the host, no sync; device values: a device-side assertion, no sync either)
            if idx.is_cuda:
                torch._assert_async(((idx >= 0) & (idx < S)).all())
            elif idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= S):
                raise rt.SitkError(f"step(indices=...): indices must lie in [0, {S}), got [{int(idx.min())}, {int(idx.max())}]")
            if getattr(self, "_prefetch", False) and not idx.is_cuda:
                if self._inp_dirty:       # load_dataset()'s index reset (main stream) must not land BEHIND this copy
                    self._side_torch.wait_event(self._ev_inp)
                    self._inp_dirty = False
                with torch.cuda.stream(self._side_torch):           # host indices: the copy rides in front of the gather on ITS stream
                    self.idx.copy_(idx.to(torch.int32), non_blocking=True)
            else:
                self.idx.copy_(idx.to(torch.int32), non_blocking=True)
                self._mark_input()
        elif x is not None:
            if self.dataset is not None:
                raise rt.SitkError("step(x, ...): a resident data set is loaded (load_dataset); select samples with "
                                   "indices=, or call unload_dataset() first")
            self.load_batch(x, target)
        segs = self._segment_fns()
        if not self.dp:
            # one GPU: nothing happens between the segments -- the whole step is ONE graph (one replay per step)
            def whole():
                for fn in segs:
                    fn()
                self._finish_backward()
                if self.loss_scaled and self.keep_grads:
                    self.fp.grad.mul_(self.gscale[1])          # .grad is read by the caller: unscaled
                if not self._defer_optimizer:                  # (a half of a SplitTrainEngine: its owner runs the pass over both halves)
                    self._optimizer()
            self._run(whole, "step")
            self.nsteps += 1
            return self.loss
        for i, fn in enumerate(segs):
            self._run(fn, i)
            if i < len(segs) - 1:
                self._allreduce_early(self.bucket_plan[i])      # final now: reduce while the remaining slices run
        if self.dp_side:
            # The side launches' weight gradients are on the SIDE stream: each bucket is reduced behind the event the library
            # recorded there at the end of ITS launch, so it runs beside the rest of the chain.  The collectives are ISSUED only
            # now, behind the host's enqueue of the whole chain: should the bucket stream share a hardware queue with the main
            # stream (ROCm maps streams of one priority onto few queues), its wait sits BEHIND the chain's launches in that queue
            # instead of in front of them (issued right after the first slice it stalled the chain for 350 us: 3.14 ms per step).
            n_side = rt.lib.sitk_overlap_side_launches(self._overlap)
            if n_side == 0:
                # no side launch was made (fewer than 2 048 tokens per rank: the large-tile weight-gradient path the side stream
                # uses does not take such a batch, every layer's gradients ran on the main stream inside the call): the early
                # buckets are final here, behind the chain -- nothing to overlap with
                for i in range(self._n_early):
                    self._allreduce_early(self.bucket_plan[i])
            elif n_side != len(self._side_groups):
                raise rt.SitkError(f"engine: {n_side} side launches, bucket plan built for {len(self._side_groups)}")
            else:
                for i in range(self._n_early):
                    last = sum(self._bucket_sizes[:i + 1]) - 1                     # the bucket is final behind its LAST side launch
                    self._allreduce_early(self.bucket_plan[i], behind_side_launch=last)
        self._run(self._finish_backward, "finish")
        # the last slice's gradients + everything `finish` wrote: from the SAME stream as the early buckets, behind the finish stage
        # -- every collective of the communicator is launched from one stream, in one order, on every rank (two streams feeding
        # one communicator would rely on the library's internal launch serialisation; nothing is gained by it: the optimizer pass
        # waits for this bucket either way)
        if self.dp_final_on == "main" and self.dp_collective == "stream":
            for lo, hi in self.bucket_plan[-1]:
                self._allreduce(lo, hi)
        else:
            self._allreduce_early(self.bucket_plan[-1])
        self._join_collectives()
        self._run(self._optimizer, "opt")
        self.nsteps += 1
        return self.loss


class SplitTrainEngine:
    """One train step as TWO concurrent half-batch steps on two streams (round 6).

    For the wide configurations (dim 384: SiT-small) every kernel of the step is a multi-round grid -- 642 GEMM workgroups on 512
    slots, LayerNorm and epilogue phases that stream while the matrix pipe idles, attention rounds with a quarter-full tail --
    and two half-batch steps side by side fill each other's gaps: measured on MI355X (tools/dual_engine.py,
    profiles/r06_split_batch.txt) SiT-small on 1280 patches, B = 32: 15.09 -> 13.87 ms per step (-8 %), on 320 patches, B = 64:
    6.82 -> 6.58 (-3.5 %); SiT-base: -0.6 .. -1.7 % (its grids have enough rounds of their own); four parts: slower than one.  The
    tiny model at BASELINE's B = 64 runs single-wave grids: nothing to fill (its own form is the side-stream step; the split loses
    4 % there) -- but above one round of workgroups it gains like the wide models (B = 128 / 256: -12 / -9 %).

    Two TrainEngines over the SAME module share one parameter buffer (FlatParams(share=...)); each owns its activations, its
    hipGraph, its stream and a gradient buffer of the same layout, and runs gather -> forward -> loss -> backward -> weight
    gradients of ITS half; the owner's optimizer pass then consumes both gradient buffers at once (sitk_*_step_dev, grad2:
    (g_a + g_b) / 2 -- each half's loss is the mean over its own samples -- and the loss (l_a + l_b) / 2), clears both and is
    the only launch on the caller's stream.  Same arithmetic as the whole-batch step up to the order of the fp32 sums over
    samples (tests/test_engine_gpu.py::test_split_engine_*).  The two streams are picked by measurement (pick_bucket_stream): the
    caller's stream is parked behind them for the length of a step, and a parked stream on the wrong hardware queue delays every
    dispatch of its pipe mates (profiles/r06_dp_streams.txt).

    Regression only, one GPU only (no process group), hipGraph replay (the wide models' form)."""

    def __init__(self, model, batch_size, **kw):
        if kw.get("task", "regression") != "regression" or kw.get("process_group") is not None:
            raise rt.SitkError("SplitTrainEngine: task='regression' on one GPU only")
        if batch_size % 2:
            raise rt.SitkError("SplitTrainEngine: the batch must be even")
        if kw.get("use_graph") is False:
            raise rt.SitkError("SplitTrainEngine: the halves replay hipGraphs (use_graph=False is the whole-batch engine's)")
        kw = dict(kw, use_graph=True, wgrad_overlap=0)
        self.device = torch.device(kw.get("device") if kw.get("device") is not None else f"cuda:{torch.cuda.current_device()}")
        self.B, self.half = batch_size, batch_size // 2
        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream(self.device)
            self.streams, self.stream_probe, self._rejected = [], [], []
            for _ in range(2):
                for attempt in range(6):
                    st, probes = pick_bucket_stream(self.device, victims=list(self.streams))
                    # ... and the other way round: the CALLER's stream is the one that sits parked while the halves run
                    rv = probe_stream(cur, main=st)
                    probes[-1]["caller_parked_ratio"] = rv["blocked_us"] / rv["free_us"]
                    if rv["blocked_us"] <= PROBE_RATIO * rv["free_us"] or attempt == 5:
                        break
                    self._rejected.append(st)         # (kept: the pool's next stream lands on another queue)
                self.streams.append(st)
                self.stream_probe.append(probes)
        self.owner = TrainEngine(model, self.half, _defer_optimizer=True, **kw)
        self.other = TrainEngine(model, self.half, _share=self.owner, **kw)
        self.parts = [self.owner, self.other]
        self.module, self.fp = self.owner.module, self.owner.fp
        self.use_graph, self.wgrad_overlap, self.keep_grads = True, 0, self.owner.keep_grads
        self.loss = self.owner.loss
        self.nsteps = 0

    @property
    def nonfinite_count(self):
        return self.owner.nonfinite_count            # (the ONE optimizer pass counts for both halves)

    def set_lr(self, lr):
        self.owner.set_lr(lr)

    def load_batch(self, x, target=None):
        h = self.half
        for i, e in enumerate(self.parts):
            e.load_batch(x[i * h:(i + 1) * h], None if target is None else target.reshape(self.B, -1)[i * h:(i + 1) * h])

    def load_dataset(self, *a, **k):
        raise rt.SitkError("SplitTrainEngine: resident data sets (load_dataset / step(indices=...)) are the whole-batch engine's")

    def _optimizer(self):
        self.owner._optimizer(other=self.other)

    def step(self, x=None, target=None, indices=None):
        """One optimisation step on the batch in the halves' static input buffers (or on x / target).  Returns the device tensor
        holding the step's loss (no host sync)."""
        if indices is not None:
            raise rt.SitkError("SplitTrainEngine: resident data sets (step(indices=...)) are the whole-batch engine's")
        if x is not None:
            self.load_batch(x, target)
        cur = torch.cuda.current_stream(self.device)
        for st, e in zip(self.streams, self.parts):
            st.wait_stream(cur)                       # behind the previous optimizer pass and this step's input copies
            with torch.cuda.stream(st):
                e.step()
        for st in self.streams:
            cur.wait_stream(st)
        if self.keep_grads:
            # kept gradients (tests): the halves unscaled theirs in place; .grad (the owner's buffer) = the batch's gradient
            self.owner.fp.grad.mul_(0.5).add_(self.other.fp.grad, alpha=0.5)
            self.owner.loss_acc.mul_(0.5).add_(self.other.loss_acc, alpha=0.5)
            self.other.fp.grad_all.zero_()
            self.owner._optimizer()                   # (zero = 0 here: the buffers are cleared at the start of the next step)
            self.loss = self.owner.loss
        else:
            self._optimizer()                         # ONE launch on the caller's stream: both gradient buffers, both cleared
        self.nsteps += 1
        return self.loss


def split_batch_by_default(dim, batch_size, task="regression", has_process_group=False, f32=False, explicit_form=False, tokens=0):
    """make_engine's rule, as a pure function of the configuration (tests/test_host_cpu.py): the split-batch form is taken where it
    was MEASURED faster by more than the spread between boxes (profiles/r06_split_batch.txt) -- regression, one GPU, an even batch,
    a 16-bit compute mode, no explicit launch-form argument, and
      * dim 384 (SiT-small): -8 .. -9 % at 1280 patches, -2 .. -3.5 % at 320;
      * dim 192 (SiT-tiny) when the batch has more tokens than ONE round of the fused kernels' 96-row workgroups (256 x 96 = 24 576):
        B = 128 / 256 / 512 on 320 patches -11 / -6 / -4 %, B = 32 on 1280 patches -8 % against the side-stream step.  At BASELINE's
        B = 64 (20 544 tokens: single-wave kernels) the side-stream step is the faster form (the split loses 4 %).
    dim 768: -0.6 .. -1.7 %, not enabled."""
    wide = dim == 384 or (dim == 192 and tokens > 256 * 96)
    return bool(wide and task == "regression" and not has_process_group and not f32 and not explicit_form
                and batch_size >= 2 and batch_size % 2 == 0)


def make_engine(model, batch_size, **kw):
    """The engine form measured fastest for the configuration: SplitTrainEngine (two concurrent half-batch steps) for a dim-384
    regression model on one GPU with an even batch -- SiT-small, BASELINE config 3 -- and for SiT-tiny batches of more than one
    round of workgroups; TrainEngine otherwise (split_batch_by_default).
    Explicit launch-form arguments (use_graph=False, wgrad_overlap, bwd_slices) select the plain engine."""
    task = kw.get("task", "regression")
    sit = model.transformer if task == "mpp" else model
    explicit = kw.get("use_graph") is False or kw.get("wgrad_overlap") is not None or kw.get("bwd_slices") is not None
    if split_batch_by_default(getattr(sit, "dim", 0), batch_size, task, kw.get("process_group") is not None,
                              rt.dtype_code(sit.compute_dtype) == rt.F32, explicit,
                              tokens=batch_size * (getattr(sit, "num_patches", 0) + 1)):
        return SplitTrainEngine(model, batch_size, **kw)
    return TrainEngine(model, batch_size, **kw)
