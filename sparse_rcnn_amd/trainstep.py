"""One data-parallel training step of the sparse path on one rank: what ``bench.py`` times and the at-size parity tests
check.  A step = index build (InputLayer rules + every rulebook; rebuilt per batch as in the reference) + forward +
backward to every parameter and the input features + all-reduce of the flat gradient buffer (RCCL / gloo, N > 1) + SGD.

Workloads (BASELINE.json configs; SURVEY.md §8d synthetic inputs):
  cfg2  configs[1]  one ~150k-voxel scene, U-Net backbone 32-64-128-256
  cfg3  configs[2] WITHOUT its RPN ("crop + mask branch only"): cfg2 + 64 synthetic boxes per scene, known before the
                    forward -> sparse ROI crop -> mask branch (maskhead.MaskBranch)
  cfg3-rpn  configs[2] with the RPN boundary INSIDE the step: backbone -> SparseToDense of the anchor level -> dense dilation
                    stack + 1x1 heads (rpn.DenseRpn: by default on THIS library's tile kernels, engine "tiles" -- a dense
                    same-convolution is a submanifold convolution on a fully active grid) -> anchors that leave the scene
                    dropped (anchor.py:103-113) -> RoiSelector (top-k + one-launch NMS) -> <= 64 boxes per scene -> sparse ROI
                    crop -> mask branch -> backward through both (rpn.py; model.py:116-240, anchor_network.py:73-124,
                    proposal_selector.py:23-89).  The boxes exist only after the heads have produced them.  Its RPN is a
                    STAND-IN, lighter than the reference's: ONE anchor level with a 2 x 32 dilation stack and 64 proposals
                    kept (BASELINE configs[2]: "~64 proposals/scene"), where scannet_config/run.py:339,525-536,609,847-853
                    builds two anchor levels with 5 x 128 / 5 x 256 stacks and keeps 256 -- `ref-crop-rpn` has that shape.
  ref-crop-rpn  the reference's own detection step as far as this path reaches: plan 32-48-64-80-96-112 on its training batch
                    (12 crops of 128 x 128 x 64, run.py:364,485-488), SparseToDense of BOTH anchor levels (stride 4: 64 ch,
                    stride 8: 80 ch), a 5 x 128 and a 5 x 256 dilation stack (run.py:525-536,609), 3 + 11 anchors per cell
                    (scannet_config/network.py:7-45) behind 1x1 heads (NOT AnchorNetworkUpsample's transposed convolutions:
                    dense, out of scope), inside-the-scene anchors only, top-1024 / NMS 0.5 / 256 kept (run.py:847-853), boxes
                    clipped to the scene (anchor.py:218-225), sparse ROI crop + mask branch.
  cfg5  configs[4]  one ~600k-voxel scene, 5-level U-Net to 512 channels
configs[3] (8 scenes data-parallel) is cfg3 with one scene per rank.
"""
from __future__ import annotations

import torch

from .dp import FlatParams, broadcast_params
from .maskhead import MaskBranch
from .synthetic import make_batch, make_boxes
from .unet import Backbone

REF_PLAN = (32, 48, 64, 80, 96, 112)      # the reference's own sparse U-Net plan, `arange * 16 + 32` (scannet_config/run.py:539-549,587-591)

WORKLOADS = {      # name -> (channels, grid, active voxels per sample, boxes per scene, BASELINE.json entry, samples per rank)
    "cfg2": ((32, 64, 128, 256), (512, 512, 256), 150_000, 0, "configs[1]", 1),
    # cfg2 with BatchNormReLU in place of every ReLU of the residual units (north_star names the operator; the reference ships it
    # off, run.py:612 `batchnorm=False`): the layer-by-layer path (the step executor does not cover batch norm), two more passes
    # over every slab and direction; under DP each rank normalises with its own scene unless modules._BatchNorm.SYNC is set
    "cfg2-bn": ((32, 64, 128, 256), (512, 512, 256), 150_000, 0, "configs[1] with BatchNormReLU units (batchnorm=True)", 1),
    "cfg3": ((32, 64, 128, 256), (512, 512, 256), 150_000, 64, "configs[2]", 1),
    "cfg3-rpn": ((32, 64, 128, 256), (512, 512, 256), 150_000, 64, "configs[2] with the RPN boundary inside the step", 1),
    "cfg5": ((32, 64, 128, 256, 512), (1024, 1024, 512), 600_000, 0, "configs[4] shape (one scene per GPU)", 1),
    # the network the reference actually trains: 6 levels 32-48-64-80-96-112 ...
    "ref": (REF_PLAN, (512, 512, 256), 150_000, 0, "configs[1] scene, the REFERENCE's own channel plan 32-48-64-80-96-112", 1),
    # ... on its own training input: 12 random crops of 128 x 128 x 64 voxels per batch (run.py:364,485-488)
    "ref-crop": (REF_PLAN, (128, 128, 64), 12_500, 0, "the reference's training batch: 12 crops of 128x128x64 voxels, "
                 "plan 32-48-64-80-96-112", 12),
    "ref-crop-rpn": (REF_PLAN, (128, 128, 64), 12_500, 256, "the reference's training batch (12 crops of 128x128x64, plan "
                     "32-48-64-80-96-112) with its RPN shape: two anchor levels, 5x128 / 5x256 dilation stacks, 256 kept", 12),
}


import os as _os

# developer switch (A/B in tools/): start the ROI batch's index build before the backbone forward (helper thread + stream)
EARLY_ROI_CUT = _os.environ.get("SCN_ROI_EARLY", "0") != "0"
# developer switches (A/B): the RPN's kernels between encoder and decoder; the prefetch thread started after the forward's kernels
RPN_BEFORE_DECODER = _os.environ.get("SCN_RPN_EARLY", "1") != "0"
# (in-process A/B, profiles/r5_ab_index_interference.txt: cfg 2 fp32 5.539 -> 5.512 ms, bf16 3.128 -> 3.094, cfg 3 neutral:
#  starting a thread costs the host ~0.1 ms exactly where the GPU's queue is shallowest, the step boundary)
LATE_PREFETCH = _os.environ.get("SCN_LATE_PREFETCH", "1") != "0"
# backward on the calling thread (torch.autograd.set_multithreading_enabled(False)): no hand-off to the device thread per step
# (A/B inside one process, profiles/r5_ab_inproc.txt: cfg 3 bf16 7.13 -> 6.51 ms per step, cfg 2 bf16 3.44 -> 3.31, fp32 neutral)
BACKWARD_INLINE = _os.environ.get("SCN_BACKWARD_INLINE", "1") != "0"


def _backward(roots, grads):
    if BACKWARD_INLINE:
        with torch.autograd.set_multithreading_enabled(False):
            torch.autograd.backward(roots, grads)
    else:
        torch.autograd.backward(roots, grads)


class SparseStepModel(torch.nn.Module):
    """Backbone (+ mask branch for cfg3) as one module, so that one flat parameter buffer covers the step."""

    def __init__(self, channels, with_mask, storage, with_rpn=False, n_boxes=64, batchnorm=False):
        """with_rpn: False | "stand-in" (cfg3-rpn: one anchor level, 2 x 32 stack) | "reference" (ref-crop-rpn: the reference's two
        anchor levels with 5 x 128 / 5 x 256 stacks, rpn.MultiLevelRpn)."""
        super().__init__()
        self.backbone = Backbone(7, channels, batchnorm=batchnorm, bf16_blocks=storage)
        self.mask = MaskBranch(channels[0], 7, bf16_blocks=storage) if with_mask else None
        self.rpn = self.roi_selector = None
        self.rpn_levels = None                 # indices of the encoder levels the RPN reads
        if with_rpn == "reference":
            from .rpn import MultiLevelRpn, RoiSelector, REF_ANCHOR_LEVELS_VOXELS
            # run.py:525-549: anchor paths on the two levels behind the in-between downsamplers (64 ch at stride 4, 80 ch at
            # stride 8 in the plan 32-48-64-80-96-112), anchor_output_channels = [128, 256], num_dilations = 5 (run.py:609)
            self.rpn_levels = (2, 3)
            self.rpn = MultiLevelRpn([(channels[2], 4, 128, REF_ANCHOR_LEVELS_VOXELS[0]),
                                      (channels[3], 8, 256, REF_ANCHOR_LEVELS_VOXELS[1])], num_dilations=5,
                                     autocast_bf16=bool(storage))
            self.roi_selector = RoiSelector(1024, n_boxes, 0.5)          # run.py:847-853: 1024 / 256 / 0.5
        elif with_rpn:               # one anchor path on the coarsest level (run.py:524: num_anchor_pathes = 1), stride 2^(L-1)
            from .rpn import DenseRpn, RoiSelector
            self.rpn_levels = (len(channels) - 1,)
            self.rpn = DenseRpn(channels[-1], stride=2 ** (len(channels) - 1), autocast_bf16=bool(storage))
            self.roi_selector = RoiSelector(1024, n_boxes, 0.5)          # run.py:848-850 with ~64 proposals kept per scene

    def run_rpn(self, interims):
        lv = [interims[i] for i in self.rpn_levels]
        return self.rpn(lv if len(lv) > 1 else lv[0])


class SceneStep:
    def __init__(self, workload="cfg2", device=None, dtype="f32", prefetch=True, seed=1, grad_seed=100, n_buckets=4,
                 target=None, channels=None, grid=None, n_boxes=None, lr=None, weighting="equal", batches_per_step=1):
        """batches_per_step: micro-batches whose gradients are accumulated before ONE all-reduce + update, each scaled by
        1 / batches_per_step -- the reference's `(loss / batches_per_step).backward()` ... `optimizer.step()`
        (ndsis/training/training.py:436,458-460; 2 or 6 with the mask head, scannet_config/run.py:377-396).  Micro-batch k
        is its own scene (seed + 1000 k); all but the last run under `FlatParams.accumulate()`.
        weighting: how the ranks' gradients are averaged -- "equal" (1 / world: balanced scenes, the benchmark) or "count"
        (each rank in proportion to its active voxels: what a loss normalised by batch-level counts gives when the
        batch is sharded one scene per rank, loss.py:401-431; the counts are summed over ranks once per step)."""
        ch, gr, tg, nb, self.baseline_entry, n_samples = WORKLOADS[workload]
        self.workload, self.dtype, self.prefetch = workload, dtype, prefetch
        # lr=None: the workload's default (reported by describe() and in bench.py's line).  1e-6, and 1e-8 with an RPN in the
        # step: the SAME synthetic gradient on 3.7 M RPN outputs every step is a steady push, not noise -- at 1e-6 the score
        # field grows 4 % per step and overflows within a bench run (profiles/r5_rpn_stats.txt); the update itself (one SGD
        # pass over the flat buffer) is the same work.  An explicit lr is used as given.
        self.lr = (1e-8 if workload.endswith("-rpn") else 1e-6) if lr is None else float(lr)
        if weighting not in ("equal", "count"):
            raise ValueError("weighting: equal | count")
        self.weighting = weighting
        self.batches_per_step = int(batches_per_step)
        if self.batches_per_step < 1:
            raise ValueError("batches_per_step >= 1")
        if self.batches_per_step > 1 and weighting == "count":
            raise ValueError("count-weighted ranks with gradient accumulation: scale each micro-batch's loss by its own count "
                             "instead (rank_weight applies to the accumulated sum when a slice is packed)")
        self._total_weight = None
        self.channels = tuple(channels or ch)
        self.grid = tuple(grid or gr)
        self.n_boxes = nb if n_boxes is None else n_boxes
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        if dtype not in ("f32", "bf16", "bf16-blocks"):
            raise ValueError("dtype: f32 | bf16 | bf16-blocks")
        storage = {"f32": False, "bf16": "all", "bf16-blocks": True}[dtype]
        self._scenes = []
        for k in range(self.batches_per_step):
            coords, feats, size, bs, splits = make_batch(n_samples, self.grid, target or tg, dup=1.15, seed=seed + 1000 * k)
            boxes = make_boxes(coords, self.n_boxes, seed=seed + 1000 * k + 2) if self.n_boxes else None
            self._scenes.append(dict(coords_cpu=coords, feats_cpu=feats, size=size, batch_size=bs, splits=splits,
                                     coords=coords.to(self.device), feats=feats.to(self.device), boxes=boxes))   # resident in HBM
        self._use_scene(0)
        torch.manual_seed(0)
        self.with_rpn = workload.endswith("-rpn")
        # ref-crop-rpn: 256 proposals per sample survive the selection (run.py:847-853); the mask network then works on the
        # <= 24 its TrainSelector draws from them (mask_network_params.selection_tuple = (24, 0, True), run.py:799-810;
        # model.py:919-1014 draws by ground-truth overlap -- out of scope: here the 24 best-scored ones)
        self.mask_boxes = 24 if workload == "ref-crop-rpn" else None
        rpn_kind = "reference" if workload == "ref-crop-rpn" else ("stand-in" if self.with_rpn else False)
        self.model = SparseStepModel(self.channels, bool(self.n_boxes), storage, rpn_kind, self.n_boxes,
                                     batchnorm=workload.endswith("-bn")).to(self.device)
        if self.with_rpn:
            self._init_rpn()
        self.flat = FlatParams(self.model, n_buckets=n_buckets)
        broadcast_params(self.flat)
        self._gen = torch.Generator(device="cpu").manual_seed(grad_seed)
        self._gys, self._gms, self._grs = {}, {}, {}
        self._gm_pool = None
        self.rpn_out = None
        self._md_next = None
        self.n_active = 0
        self.n_roi_rows = 0
        self.out = self.logits = self.fin = None

    # ------------------------------------------------------------------------------------------------------------
    def _use_scene(self, k):
        sc = self._scenes[k]
        self.coords_cpu, self.feats_cpu, self.size, self.batch_size, self.splits = (
            sc["coords_cpu"], sc["feats_cpu"], sc["size"], sc["batch_size"], sc["splits"])
        self.coords, self.feats, self.boxes = sc["coords"], sc["feats"], sc["boxes"]
        self._k = k

    def _scene_shape(self):
        return tuple(float(v) for v in self.size)

    def _init_rpn(self):
        """Random-init heads give near-constant scores; the synthetic RPN gets a head whose scores spread (so that top-k and
        NMS have something to decide) -- seeded, the same on every rank."""
        g = torch.Generator().manual_seed(1234)
        rpn = self.model.rpn
        with torch.no_grad():
            for h in ([r.head for r in rpn.levels] if hasattr(rpn, "levels") else [rpn.head]):
                w = torch.randn(h.weight.shape, generator=g) * 0.02              # box deltas: boxes stay near their anchors
                w[6::7] = torch.randn(w[6::7].shape, generator=g) * 0.5          # channel a*7+6 = the score of anchor a
                h.weight.copy_(w.to(h.weight.device))
                h.bias.zero_()

    def _take_index(self, k):
        """The index structures of micro-batch k if a helper thread built them (else None: the forward builds them)."""
        md = self._md_next.result() if self._md_next is not None else None
        self._md_next = None
        return md

    def _start_prefetch(self, k):
        if self.prefetch and self._md_next is None:
            nx = self._scenes[(k + 1) % self.batches_per_step]
            self._md_next = self.model.backbone.prefetch_in_thread(nx["coords"], nx["size"], nx["batch_size"])

    def upstream_grads(self, k=0):
        """(dY of the backbone output, dY of the mask logits or None) of micro-batch k, BEFORE the 1 / batches_per_step scale."""
        return self._gys.get(k), self._gms.get(k)

    def forward_backward(self, k=0, zero=True):
        """Index build + forward + backward of micro-batch k (no collective, no update); the upstream gradients are scaled
        by 1 / batches_per_step.  zero: drop the gradients first (the first micro-batch of a step).
        Keeps .out / .logits / .fin for checks."""
        m = self.model
        # (the gradients are dropped right before backward, not here: at the step boundary the GPU's queue is empty, and 156
        #  attribute stores are 40-50 us the first forward kernels would wait for)
        if k != self._k:
            self._use_scene(k)
        scale = 1.0 / self.batches_per_step
        fin = self.feats.detach().requires_grad_()
        md = self._take_index(k)
        # the index structures of the NEXT batch depend on its coordinates only (a data loader's output): a helper thread
        # builds them on the high-priority index stream while this batch runs; every step contains one complete build
        if not LATE_PREFETCH:
            self._start_prefetch(k)
        # cfg3-rpn: the RPN reads the ENCODER outputs only (model.py:141-160), so its heads, top-k and NMS are queued between
        # encoder and decoder, and the one host wait of the selection falls while the decoder's kernels run
        rpn_state = {}

        def rpn_after_encoder(interims):
            rpn_bbox, rpn_score, anchors = m.run_rpn(interims)
            rpn_state["out"] = (rpn_bbox, rpn_score, anchors,
                                m.roi_selector.start(rpn_bbox, rpn_score, anchors, self._scene_shape()))
        hook = rpn_after_encoder if (self.with_rpn and RPN_BEFORE_DECODER) else None
        # cfg3: the ROI crop's selection and the ROI batch's index structures depend on coordinates and boxes only -- the
        # boxes of a step are known before its backbone runs (here: synthetic; in the reference: the RPN's proposals of
        # the same forward, so this applies to the mask branch's SECOND use of a scene, e.g. evaluation on cached proposals)
        cut = None
        if m.mask is not None and EARLY_ROI_CUT:
            cut = m.mask.prepare_cut(self.coords, self.size, self.boxes)      # (resident int64 coords: no dependency on md)
        out = m.backbone(self.coords, fin, self.size, self.batch_size, metadata=md, after_encoder=hook)
        self._start_prefetch(k)      # (LATE_PREFETCH: the helper thread is started once this batch's forward kernels are queued)
        gy = self._gys.get(k)
        if gy is None or gy.shape != out.features.shape:
            gy = self._gys[k] = torch.randn(out.features.shape, generator=self._gen).to(self.device)   # upstream grad dY ~ N(0,1)
            self.n_active = sum(g.shape[0] for g in self._gys.values())
        gys = gy if scale == 1.0 else gy * scale
        if self.weighting == "count":             # before backward: the bucketed path scales slices as it packs them
            import torch.distributed as dist
            self.flat.rank_weight = float(out.features.shape[0])
            tot = torch.tensor([self.flat.rank_weight], dtype=torch.float64)
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                tot = tot.to(self.device if dist.get_backend() == "nccl" else "cpu")
                dist.all_reduce(tot)
            self._total_weight = float(tot.item())
        if m.mask is None:
            if zero:
                self.flat.zero_grad()
            _backward([out.features], [gys])
            logits = None
        else:
            scene = (self.coords, fin, self.size, self.batch_size, self.splits)
            boxes, roots, root_grads = self.boxes, [out.features], [gys]
            if self.with_rpn:
                # configs[2] as written (model.py:141-160): the proposals are this forward's -- dense heads on the coarsest
                # encoder level, top-k + NMS on the device; the RPN losses' gradients arrive at rpn_bbox / rpn_score
                if "out" not in rpn_state:
                    rpn_after_encoder(m.backbone.unet.interims)
                rpn_bbox, rpn_score, anchors, sel_state = rpn_state["out"]
                roi_score, boxes, roi_index = m.roi_selector.finish(sel_state)
                self.rpn_out = (rpn_bbox, rpn_score, anchors, roi_score, boxes, roi_index)
                if self.mask_boxes is not None:       # (the reference's mask head trains on <= 24 selected proposals per sample)
                    boxes = [b[:self.mask_boxes] for b in boxes]
                gr = self._grs.get(k)
                if gr is None or gr[0].shape != rpn_bbox.shape:
                    gr = self._grs[k] = tuple((torch.randn(t.shape, generator=self._gen) * 1e-3).to(self.device)
                                              for t in (rpn_bbox, rpn_score))
                roots += [rpn_bbox, rpn_score]
                root_grads += [g if scale == 1.0 else g * scale for g in gr]
            logits, selection = m.mask(scene, out, boxes, prepared_cut=cut)
            gm = self._gms.get(k)
            if gm is None or gm.shape != logits.shape:
                if self.with_rpn:
                    # the proposals -- and with them the number of cropped points -- change from step to step: dM is a slice
                    # of one device-resident pool (drawing 2 M normals on the host per step would be timed as part of it)
                    pool = self._gm_pool
                    if pool is None or pool.shape[0] < logits.shape[0] or pool.shape[1:] != logits.shape[1:]:
                        rows = max(2 * logits.shape[0], 1 << 18)
                        pool = self._gm_pool = torch.randn((rows,) + tuple(logits.shape[1:]), generator=self._gen).to(self.device)
                    gm = self._gms[k] = pool[:logits.shape[0]]
                else:
                    gm = self._gms[k] = torch.randn(logits.shape, generator=self._gen).to(self.device)
                self.n_roi_rows = sum(g.shape[0] for g in self._gms.values())
            if logits.requires_grad and logits.shape[0]:
                roots.append(logits)
                root_grads.append(gm if scale == 1.0 else gm * scale)
            # (an empty crop -- no proposal caught a point: the mask branch contributes nothing on this rank)
            if zero:
                self.flat.zero_grad()
            _backward(roots, root_grads)
        self.out, self.logits, self.fin = out, logits, fin

    def forward_only(self, k=0):
        """Evaluation forward of micro-batch k under torch.no_grad() -- what the reference's `eval_model`
        (ndsis/training/training.py:244-304) and `SparseMaskPredictor` (model.py:826-882) run: index build + backbone
        (+ ROI crop + mask branch), no graph, the executor's forward-only slab plan (executor._lean_layout), no
        backward-data weight images.  Same bits as the training forward.  -> (backbone output tensor, mask logits | None)"""
        m = self.model
        if k != self._k:
            self._use_scene(k)
        md = self._take_index(k)
        self._start_prefetch(k)
        with torch.no_grad():
            out = m.backbone(self.coords, self.feats, self.size, self.batch_size, metadata=md)
            logits = None
            if m.mask is not None:
                scene = (self.coords, self.feats, self.size, self.batch_size, self.splits)
                boxes = self.boxes
                if self.with_rpn:
                    rpn_bbox, rpn_score, anchors = m.run_rpn(m.backbone.unet.interims)
                    _, boxes, _ = m.roi_selector(rpn_bbox, rpn_score, anchors, self._scene_shape())
                    if self.mask_boxes is not None:
                        boxes = [b[:self.mask_boxes] for b in boxes]
                logits, _ = m.mask(scene, out, boxes)
        return out, logits

    def step(self):
        n = self.batches_per_step
        for k in range(n - 1):                    # training.py:436: (loss / batches_per_step).backward(), no update yet
            with self.flat.accumulate():
                self.forward_backward(k, zero=(k == 0))
        self.forward_backward(n - 1, zero=(n == 1))       # the last micro-batch: bucket hooks armed, slices go out
        if self.weighting == "count":
            self.flat.all_reduce_mean(total_weight=self._total_weight)
            self.flat.sgd_step(self.lr)
            return
        self.flat.step_single_rank(self.lr)      # = all_reduce_mean + sgd_step; one rank: no packing into the flat bucket

    def finish(self):
        """Join the index build started by the last step (it belongs to the timed region)."""
        if self._md_next is not None:
            self._md_next.result()
            self._md_next = None

    def describe(self):
        s = (f"BASELINE {self.baseline_entry}: {self.batch_size} synthetic ScanNet-shaped sample(s) per GPU, {self.n_active} "
             f"active voxels (grid {self.grid[0]}x{self.grid[1]}x{self.grid[2]}, 1.15 points/voxel), U-Net "
             + "-".join(map(str, self.channels)) + ", 2 pre-act residual blocks/level, 2^3/2 conv+deconv"
             + (", BatchNormReLU in the residual units (training mode, fp64 statistics; layer-by-layer path)"
                if self.workload.endswith("-bn") else ""))
        if self.n_boxes and self.with_rpn:
            r = self.model.rpn
            sel = (f"-> anchors that leave the scene dropped (anchor.py:103-113) -> sigmoid, top-1024, boxes clipped to the scene, "
                   f"one-launch NMS 0.5, <= {self.n_boxes} boxes/sample" + (f", the {self.mask_boxes} best of them per sample" if self.mask_boxes else "")
                   + f" -> sparse ROI crop ({self.n_roi_rows} cropped points) -> mask "
                   "branch (SubM1 + 2 units @16, internal U-Net 23-32-48-64, Linear 23-32-18); backward from the backbone "
                   "output, rpn_bbox, rpn_score and the mask logits")
            if hasattr(r, "levels"):
                eng = r.levels[0].engine or r.levels[0].ENGINE
                s += ("; + the REFERENCE's RPN shape INSIDE the step (run.py:525-536,609,847-853): " + " + ".join(
                    f"SparseToDense of the stride-{l.stride} level ({l.channels} ch) -> dense dilation stack {l.channels}"
                    + f"-{l.width}" * (len(l.stack) // 2) + f" (3^3) + 1x1 head ({l.n_anchors} anchors/cell)" for l in r.levels)
                    + f", dense layers on engine '{eng}' "
                    + ("(this library's tile kernels on a fully active grid)" if eng == "tiles" else "(torch / MIOpen conv3d)")
                    + "; 1x1 heads instead of AnchorNetworkUpsample's transposed convolutions (dense, out of scope) " + sel)
            else:
                eng = r.engine or r.ENGINE
                s += (f"; + RPN boundary INSIDE the step, a STAND-IN lighter than the reference's (one anchor level, 2 x {r.width} "
                      f"stack, {self.n_boxes} kept; the reference: two levels, 5 x 128 / 5 x 256, 256 kept -- `ref-crop-rpn`): "
                      f"SparseToDense of the stride-{r.stride} level ({r.channels} ch) -> dense dilation stack {r.channels}"
                      + f"-{r.width}" * (len(r.stack) // 2) + f" (3^3, engine '{eng}': "
                      + ("this library's tile kernels on a fully active grid" if eng == "tiles" else "torch / MIOpen conv3d")
                      + f") + 1x1 head ({r.n_anchors} anchors/cell) " + sel)
        elif self.n_boxes:
            s += (f"; + {self.n_boxes} fp32 boxes/scene (edges 8-96 voxels) -> sparse ROI crop ({self.n_roi_rows} cropped "
                  "points) -> mask branch (SubM1 + 2 units @16, internal U-Net 23-32-48-64, Linear 23-32-18); CROP + MASK "
                  "BRANCH ONLY: the boxes are synthetic and known before the forward (no RPN in this step; "
                  "--workload cfg3-rpn has it)")
        if self.batches_per_step > 1:
            s += (f"; {self.batches_per_step} micro-batches (scenes) accumulated per optimizer step (training.py:436,458-460), "
                  "voxels = all of them")
        s += (f"; step = rulebooks + fwd + bwd (+ grad all-reduce) + plain SGD on the flat parameter buffer, lr {self.lr:g} "
              "(the reference trains with Adam, scannet_config/run.py:1449: three more passes over the buffer)")
        if self.prefetch:
            s += "; rulebooks of batch i+1 built on a helper thread during batch i"
        return s
