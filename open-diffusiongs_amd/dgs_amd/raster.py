"""Torch-tensor front end of the rasterizer C ABI (include/dgs_raster.h).

`RasterBackend.rasterize_gaussians / rasterize_gaussians_backward / mark_visible` have exactly the
argument lists and return tuples of the reference's pybind module `_C`
(/root/reference/submodules/diff-gaussian-rasterization/rasterize_points.h:18-66), so
diff_gaussian_rasterization/__init__.py can stay a verbatim mirror of the reference binding.
PyTorch only provides device memory and the current HIP stream here.
"""
import ctypes
import os
import time

import torch

from . import _native


def _ptr(t):
    """Device pointer of an optional tensor: empty tensors mean 'absent' (reference: torch.Tensor([]))."""
    if t is None or t.numel() == 0:
        return None
    return ctypes.c_void_p(t.data_ptr())


def _prep(t, device):
    if t is None or t.numel() == 0:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    if t.device != device:
        t = t.to(device)
    return t.contiguous()


class _AsyncPlan:
    """What the backend knows about the calls of ONE shape (P, W, H, V, views per set, device): the instance statistics the device
    reported for the calls so far.  They size the next call's binning buffer and pick its ordering form, so that the call itself
    reads nothing back (the reference blocks on `num_rendered` in every forward, rasterizer_impl.cu:281, and can therefore never
    fail on an instance count: rasterize_points.cu:27-33,76-78 resize by callback).  The same guarantee here, without the block:

    capacity  max(2 x the most instances any call of this shape produced,  min(worst case, budget)) -- the worst case is V x T x P
              (every Gaussian in every tile); forward-only renders (`Renderer.forward`) may spend `budget_bytes` on the buffer
              (torch memory: a cached block after the first call; only the part a scene fills is ever touched), which at the
              object model's 256^2 shape IS the worst case (3.2 GB of 288): such a plan cannot overflow and never waits.
    at risk   a plan whose capacity is below the worst case.  Its calls are VERIFIED: the host waits for the four statistics words
              (stored by the call's second kernel, long before the blend) and, if the scene outgrew the buffer, renders it again
              with a buffer sized for it -- on the same stream, into the same outputs, before anything else was enqueued.  What a
              caller sees is the reference's behaviour: a correct image, always; what it pays is the host's run-ahead at that
              point (the device does not idle: the rest of the render is already enqueued).  A call that is being CAPTURED cannot
              wait: the graph's owner verifies each replay instead (dgs_amd/graph.py).
    form      dgs_raster_binning_form() of the latest statistics that have arrived (any form is correct for any scene -- the lists
              are bit-identical -- a stale one only costs time); `longest` = the longest tile list seen (sizes the LDS sort's LDS;
              a longer list takes that kernel's chunked path).
    Statistics travel device -> pinned host words behind each call; `poll` looks at the ones that have arrived."""

    MARGIN = 2.0
    SLOTS = 64                     # pinned host words for the statistics of that many calls in flight
    BUDGET_BYTES = int(float(os.environ.get("DGS_RASTER_PLAN_GIB", "4")) * 2 ** 30)   # forward-only plans: floor of the binning buffer

    def __init__(self, worst=0, bytes_per_instance=12):
        self.capacity, self.form, self.longest, self.seen_max = 0, 0, 0, 0
        self.worst = int(worst)    # V x T x P: no scene of this shape has more instances
        self.bytes_per_instance = int(bytes_per_instance)
        self.pending = []          # (event or None, host int32[4], capacity the call ran with)
        self.overflowed = None     # (instances, capacity) of an UNVERIFIED call that did not fit (a capture without an owner), until reported
        self.calls = {"sync": 0, "async": 0, "healed": 0}
        self.host = None           # pinned int32[SLOTS + 1, 4]: allocated ONCE, outside any stream capture (a pinned allocation is not a
        self.slot = 0              # stream operation: inside a capture it invalidates the capture); row SLOTS: captures without an owner

    def capacity_for(self, forward_only):
        """Instances the next call's binning buffer holds.  Forward-only callers get the budget's floor on top of the history; a call
        whose state feeds a backward does not (the deterministic backward's scratch is 36 bytes per SLOT)."""
        cap = self.capacity
        if forward_only and cap > 0 and self.worst > 0:
            cap = max(cap, min(self.worst, self.BUDGET_BYTES // self.bytes_per_instance))
        return min(cap, self.worst) if self.worst > 0 else cap

    def at_risk(self, capacity):
        return self.worst <= 0 or capacity < self.worst

    def host_words(self, for_graph):
        if self.host is None:
            self.host = torch.empty(self.SLOTS + 1, 4, dtype=torch.int32, pin_memory=True)
        if for_graph:
            return self.host[self.SLOTS]
        if len(self.pending) >= self.SLOTS:          # every slot holds an unread result: wait for the oldest
            ev = self.pending[0][0]
            if ev is not None:
                ev.synchronize()
        w = self.host[self.slot]
        self.slot = (self.slot + 1) % self.SLOTS
        return w

    @staticmethod
    def _grid(n):
        """n rounded up to a grid of eight steps per power of two.  A training run's instance count creeps by a few thousand every
        step; a capacity that followed it exactly asked the allocator for a slightly LARGER multi-gigabyte buffer every step -- no cached
        block ever fits, each step pays a fresh hipMalloc (1.2 s per step with the 45 GB scratch of the deterministic backward,
        profiles/r05_train_det_alloc.txt) and the freed blocks pile up in the cache."""
        n = max(int(n), 1024)
        step = max(1 << (n.bit_length() - 1), 8) >> 3
        return -(-n // step) * step

    def note(self, lib, n, longest, P, W, H, V):
        self.seen_max = max(self.seen_max, int(n))
        self.longest = max(self.longest, int(longest))
        self.capacity = max(self.capacity, self._grid(self.MARGIN * self.seen_max + 1024))
        self.form = int(lib.dgs_raster_binning_form(0, int(n), int(longest), P, W, H, V))

    def poll(self, lib, P, W, H, V, wait=False):
        keep = []
        if len(self.pending) >= self.SLOTS:
            wait = True                                  # the ring of host words is full
        for ev, host, cap in self.pending:
            if ev is not None and not wait and not ev.query():
                keep.append((ev, host, cap))
                continue
            if ev is not None and wait:
                ev.synchronize()
            n, status, longest = int(host[0]) & 0xFFFFFFFF, int(host[1]), int(host[2]) & 0xFFFFFFFF
            self.note(lib, n, longest, P, W, H, V)
            if status == _native.DGS_ERR_BINNING_OVERFLOW:
                self.overflowed = (n, cap, longest)
            elif status != 0:
                raise RuntimeError(f"dgs rasterizer: an earlier asynchronous call failed on the device: {_native.status_string(lib, status)} (status {status})")
        self.pending = keep
        if self.overflowed is not None:
            n, cap, longest = self.overflowed
            self.overflowed = None
            raise RuntimeError(f"dgs rasterizer: an earlier UNVERIFIED asynchronous render (a call captured into a graph that nobody "
                               f"verifies: use dgs_amd.graph.GraphedForward) outgrew its binning buffer -- {n} instances for {cap}, longest "
                               f"tile list {longest} -- that call's image is NaN.  The plan now holds capacity {self.capacity}, form {self.form}")


class RasterBackend:
    def __init__(self, lib=None, exact_exp=None):
        self.lib = lib if lib is not None else _native.lib()
        self._plans = {}             # (P, W, H, V, views_per_set, device) -> _AsyncPlan
        self.verify_wait_s = 0.0     # host seconds spent in the verify wait of plans at risk (bench.py splits its host-enqueue figure with it)
        # exponential of the blend loops (dgs_raster.h `exact_exp`): False = compensated v_exp_f32 + cut-off guard band (product default, <= 1.3 ulp), True = the
        # fixed IEEE sequence the CPU oracle restates (floats bit-identical with the oracle: what the bit-exact parity tests select)
        self.exact_exp = bool(int(os.environ.get("DGS_RASTER_EXACT_EXP", "0") or 0)) if exact_exp is None else bool(exact_exp)
        # backward without floating-point atomics (dgs_raster.h `scratch`; DGS_RASTER_DETERMINISTIC=1 or `backend.deterministic = True`):
        # bit-reproducible gradients for 36 bytes of scratch per instance slot and, in dense scenes, time; if
        # the scratch would exceed `deterministic_budget` bytes the atomic form runs and `last_backward_deterministic` says so
        self.deterministic = os.environ.get("DGS_RASTER_DETERMINISTIC", "0") not in ("", "0")
        self.deterministic_budget = int(float(os.environ.get("DGS_RASTER_DETERMINISTIC_GIB", "64")) * 2 ** 30)
        self.last_backward_deterministic = None
        self.last_async_stats = None     # device int32[4] of the latest NON-planned asynchronous call (a caller with its own capacity)
        self._capture_log, self._capture_rows = [], []

    # -- helpers ---------------------------------------------------------------------------
    @staticmethod
    def _stream(device):
        if device.type == "cuda":
            return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        return None

    def _check(self, rc):
        if rc != 0:
            msg = _native.status_string(self.lib, rc)
            raise RuntimeError(f"dgs rasterizer: {msg} (status {rc})")

    @staticmethod
    def _allocator(holder, key, device):
        def cb(nbytes, _user):
            holder[key] = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            return holder[key].data_ptr()
        return _native.ALLOC_FN(cb)

    # -- _C.rasterize_gaussians -----------------------------------------------------------
    def rasterize_gaussians(self, background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                            viewmatrix, projmatrix, tanfovx, tanfovy, image_height, image_width, sh, degree, campos,
                            prefiltered, debug, raw_activations=False):
        if means3D.dim() != 2 or means3D.size(1) != 3:
            raise RuntimeError("means3D must have dimensions (num_points, 3)")   # rasterize_points.cu:57-59
        out = self.forward_views(background, means3D[None], colors, opacity, scales, rotations, scale_modifier,
                                 cov3D_precomp, viewmatrix.reshape(1, 4, 4), projmatrix.reshape(1, 4, 4),
                                 campos.reshape(1, 3), None, float(tanfovx), float(tanfovy), image_height, image_width,
                                 sh, degree, prefiltered, debug, views_per_set=1, raw_activations=raw_activations)
        num_rendered, color, radii, geom, binning, img = out
        return num_rendered, color[0], radii[0], geom, binning, img

    def plan_for(self, P, W, H, V, views_per_set, device):
        key = (int(P), int(W), int(H), int(V), int(views_per_set), str(device))
        plan = self._plans.get(key)
        if plan is None:
            tiles = ((int(W) + 15) // 16) * ((int(H) + 15) // 16)
            per = max(1, int(self.lib.dgs_raster_binning_bytes(1 << 20)) >> 20)
            plan = self._plans[key] = _AsyncPlan(worst=int(V) * tiles * int(P), bytes_per_instance=per)
        return plan

    def check_async(self, wait=True):
        """Look at the statistics of every asynchronous call so far (wait=True: block until their copies have arrived) and raise if
        one of them failed on the device -- the place for a caller to turn a NaN image into an exception at a moment of its choosing
        (end of a sampling loop, end of a training step)."""
        for (P, W, H, V, _vps, _dev), plan in self._plans.items():
            plan.poll(self.lib, P, W, H, V, wait=wait)

    def forward_views(self, background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                      viewmatrix, projmatrix, campos, tanfov, tanfovx, tanfovy, image_height, image_width, sh, degree,
                      prefiltered, debug, views_per_set=1, raw_activations=False, binning_capacity=0, planned=False, forward_only=False):
        """Batched entry: means3D [S,P,3] (other per-Gaussian inputs [S,P,...]); viewmatrix/projmatrix [V,4,4];
        campos [V,3]; tanfov optional [V,2] tensor.  Returns (num_rendered, color[V,3,H,W], radii[V,P], geom, binning, img).
        planned=True (the product's render path: Renderer.forward): no host synchronisation -- the binning capacity and the ordering
        form come from the `_AsyncPlan` of this shape; the FIRST call of a shape runs the synchronous form to learn them.  The
        returned `num_rendered` is then the capacity the binning buffer was carved with (what the backward takes); a scene that
        outgrows the plan is rendered again before the call returns (`_AsyncPlan`), never reported as NaN.  forward_only=True: no
        backward will take this call's state, so the buffer may be sized for the worst case within the plan's budget."""
        device = means3D.device
        S, P = int(means3D.shape[0]), int(means3D.shape[1])
        V = int(viewmatrix.shape[0])
        H, W = int(image_height), int(image_width)
        means3D = _prep(means3D, device)
        keep = dict(bg=_prep(background, device), means=means3D, colors=_prep(colors, device), op=_prep(opacity, device),
                    scales=_prep(scales, device), rots=_prep(rotations, device), cov=_prep(cov3D_precomp, device),
                    vm=_prep(viewmatrix, device), pm=_prep(projmatrix, device), cam=_prep(campos, device),
                    sh=_prep(sh, device), tanfov=_prep(tanfov, device))
        M = 0
        if keep["sh"] is not None:
            M = int(keep["sh"].shape[-2])
        # rasterize_points.cu:64-65 allocate zeros; here the kernels write every pixel and every radius (P == 0: nothing runs)
        alloc = torch.zeros if P == 0 else torch.empty
        out_color = alloc((V, 3, H, W), dtype=torch.float32, device=device)
        radii = alloc((V, P), dtype=torch.int32, device=device)
        holder = {"geom": torch.empty(0, dtype=torch.uint8, device=device),
                  "binning": torch.empty(0, dtype=torch.uint8, device=device),
                  "img": torch.empty(0, dtype=torch.uint8, device=device)}
        if P == 0:
            return 0, out_color, radii, holder["geom"], holder["binning"], holder["img"]
        a = _native.DgsRasterForwardArgs()
        a.P, a.D, a.M, a.width, a.height, a.V, a.views_per_set = P, int(degree), M, W, H, V, int(views_per_set)
        a.background = _ptr(keep["bg"]); a.means3D = _ptr(means3D); a.shs = _ptr(keep["sh"])
        a.colors_precomp = _ptr(keep["colors"]); a.opacities = _ptr(keep["op"]); a.scales = _ptr(keep["scales"])
        a.rotations = _ptr(keep["rots"]); a.cov3D_precomp = _ptr(keep["cov"]); a.viewmatrix = _ptr(keep["vm"])
        a.projmatrix = _ptr(keep["pm"]); a.campos = _ptr(keep["cam"]); a.tanfov = _ptr(keep["tanfov"])
        a.tanfovx, a.tanfovy, a.scale_modifier = float(tanfovx), float(tanfovy), float(scale_modifier)
        a.prefiltered, a.debug, a.raw_activations = int(bool(prefiltered)), int(bool(debug)), int(bool(raw_activations))
        a.out_color = ctypes.c_void_p(out_color.data_ptr())
        a.radii = ctypes.c_void_p(radii.data_ptr())
        cbs = [self._allocator(holder, k, device) for k in ("geom", "img", "binning")]
        a.geom_alloc, a.img_alloc, a.binning_alloc = cbs
        a.binning_form = int(os.environ.get("DGS_RASTER_BIN", "0") or 0)     # tests / measurement only (dgs_raster.h)
        a.exact_exp = int(self.exact_exp)
        plan = self.plan_for(P, W, H, V, views_per_set, device) if planned else None
        capturing = device.type == "cuda" and torch.cuda.is_current_stream_capturing()
        forced_form = a.binning_form
        if plan is not None and not capturing:
            plan.poll(self.lib, P, W, H, V)
        for _attempt in range(4):
            if plan is not None:
                binning_capacity = plan.capacity_for(forward_only)   # 0 before the first call of the shape: the synchronous form, which reports N
                if binning_capacity > 0 and forced_form == 0:
                    a.binning_form, a.longest_hint = plan.form, plan.longest
                if capturing and binning_capacity <= 0:
                    raise RuntimeError("dgs rasterizer: the first render of a shape synchronises (it learns the binning capacity); run the step "
                                       "once before capturing it in a graph")
            a.binning_capacity = int(binning_capacity)
            ndev = host = pre = None
            if binning_capacity > 0:
                if plan is not None and device.type == "cuda":
                    # statistics -> pinned host words, stored by the call's scan kernel (device-mapped pinned memory: no copy node in a
                    # captured call); looked at below (a plan at risk), by a later call's poll(), or by the owner of the graph
                    host = self._capture_row() if capturing else None
                    if host is None:
                        host = plan.host_words(for_graph=capturing)
                    a.num_rendered_host = ctypes.c_void_p(host.data_ptr())
                    if not capturing:
                        host[3] = 0                       # the call's scan kernel sets it behind the other three words
                        if plan.at_risk(binning_capacity):
                            pre = torch.cuda.Event()
                            pre.record(torch.cuda.current_stream(device))
                else:
                    ndev = torch.empty(4, dtype=torch.int32, device=device)
                    a.num_rendered_dev = ctypes.c_void_p(ndev.data_ptr())
            rc = self.lib.dgs_raster_forward(ctypes.byref(a), self._stream(device))
            self._check(rc)
            if binning_capacity <= 0:                 # synchronous form: the statistics are in the argument block
                if plan is not None:
                    plan.calls["sync"] += 1
                    plan.note(self.lib, int(a.num_rendered), int(a.longest_list), P, W, H, V)
                    if device.type == "cuda":
                        plan.host_words(for_graph=True)   # the pinned words exist before anybody captures a call of this shape
                return int(a.num_rendered), out_color, radii, holder["geom"], holder["binning"], holder["img"]
            if plan is None:                          # a caller with a capacity of its own: the four words are its business
                self.last_async_stats = ndev
                return int(binning_capacity), out_color, radii, holder["geom"], holder["binning"], holder["img"]
            plan.calls["async"] += 1
            if capturing:
                # rewritten by every replay; the graph's owner (dgs_amd/graph.py) verifies each replay of a plan at risk
                plan.graph_stats = host
                self._capture_log.append((plan, host, int(binning_capacity), (P, W, H, V)))
                break
            if host is not None:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(device))
            else:
                ev, host = None, ndev
            if not plan.at_risk(binning_capacity):
                plan.pending.append((ev, host if ev is not None else ndev.clone(), int(binning_capacity)))
                break
            # a plan at risk: verify before anything else is enqueued (class docstring).  The words are stored by the call's second
            # kernel, word [3] last: the host first sleeps until the stream has reached this call (`pre`), then polls that word -- it
            # turns up ~0.1 ms into the call, while the blend kernels are still to run, so the device has work while the host resumes
            # (waiting for the END of the call cost the rasterizer microbenchmark +0.15 ms per forward + backward:
            # profiles/r05_raster256_*.log against r04's).  A word that does not turn up within two seconds: the event.
            if ev is not None:
                w0 = time.perf_counter()
                if pre is not None:
                    pre.synchronize()
                deadline = time.perf_counter() + 2.0
                while int(host[3]) == 0 and time.perf_counter() < deadline:
                    pass
                if int(host[3]) == 0:
                    ev.synchronize()
                self.verify_wait_s += time.perf_counter() - w0       # host time spent waiting for the device to reach / start this call
            n, status, longest = int(host[0]) & 0xFFFFFFFF, int(host[1]), int(host[2]) & 0xFFFFFFFF
            plan.note(self.lib, n, longest, P, W, H, V)
            if status == 0:
                break
            if status != _native.DGS_ERR_BINNING_OVERFLOW:
                raise RuntimeError(f"dgs rasterizer: {_native.status_string(self.lib, status)} (status {status})")
            plan.calls["healed"] += 1                 # the buffer was too small: again, now sized for this scene (same stream, same outputs)
        else:
            raise RuntimeError("dgs rasterizer: a render kept outgrowing its binning buffer")
        return int(binning_capacity), out_color, radii, holder["geom"], holder["binning"], holder["img"]

    # -- captured calls: the owner of a graph hands out the pinned rows its calls report into and learns which plans they used ----
    def begin_capture_log(self, rows=None):
        """Called by the owner of a graph before it captures: `rows` = pinned int32[k, 4] allocated outside the capture (one row per
        rasterizer call of the captured sequence; without it the plan's shared row is used).  `end_capture_log()` returns
        [(plan, row, capacity, (P, W, H, V))] of the calls captured since."""
        self._capture_log = []
        self._capture_rows = list(rows) if rows is not None else []

    def end_capture_log(self):
        log, self._capture_log, self._capture_rows = self._capture_log, [], []
        return log

    def _capture_row(self):
        return self._capture_rows.pop(0) if self._capture_rows else None

    # -- _C.rasterize_gaussians_backward ----------------------------------------------------
    def rasterize_gaussians_backward(self, background, means3D, radii, colors, scales, rotations, scale_modifier,
                                     cov3D_precomp, viewmatrix, projmatrix, tanfovx, tanfovy, dL_dout_color, sh, degree,
                                     campos, geomBuffer, R, binningBuffer, imageBuffer, debug):
        """rasterize_points.cu:117-196: -> (dL_dmeans2D[P,3], dL_dcolors[P,3], dL_dopacity[P,1], dL_dmeans3D[P,3],
        dL_dcov3D[P,6], dL_dsh[P,M,3], dL_dscales[P,3], dL_drotations[P,4])."""
        H, W = int(dL_dout_color.shape[-2]), int(dL_dout_color.shape[-1])
        g = self.backward_views(background, means3D[None], radii[None], colors, None, scales, rotations, scale_modifier,
                                cov3D_precomp, viewmatrix.reshape(1, 4, 4), projmatrix.reshape(1, 4, 4), campos.reshape(1, 3),
                                None, float(tanfovx), float(tanfovy), dL_dout_color.reshape(1, 3, H, W), sh, degree,
                                geomBuffer, R, binningBuffer, imageBuffer, debug, views_per_set=1)
        P = int(means3D.shape[0])
        M = int(sh.shape[-2]) if sh is not None and sh.numel() else 0
        return (g["means2D"][0], g["colors"].reshape(-1, 3)[:P], g["opacity"].reshape(P, 1), g["means3D"][0], g["cov3D"][0],
                g["sh"].reshape(P, M, 3), g["scales"].reshape(-1, 3)[:P] if g["scales"] is not None else torch.zeros((P, 3), device=means3D.device),
                g["rotations"].reshape(-1, 4)[:P] if g["rotations"] is not None else torch.zeros((P, 4), device=means3D.device))

    def backward_views(self, background, means3D, radii, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                       viewmatrix, projmatrix, campos, tanfov, tanfovx, tanfovy, dL_dpix, sh, degree, geom, num_rendered,
                       binning, img, debug, views_per_set=1, raw_activations=False):
        """Batched backward: means3D [S,P,3], radii [V,P], dL_dpix [V,3,H,W] -> dict of gradients
        (means2D [V,P,3], cov3D [V,P,6] per view; means3D, opacity, scales, rotations, sh (and colors when precomputed)
        summed over the views of each set)."""
        device = means3D.device
        S, P = int(means3D.shape[0]), int(means3D.shape[1])
        V = int(viewmatrix.shape[0])
        H, W = int(dL_dpix.shape[-2]), int(dL_dpix.shape[-1])
        keep = dict(bg=_prep(background, device), means=_prep(means3D, device), colors=_prep(colors, device),
                    op=_prep(opacity, device), scales=_prep(scales, device), rots=_prep(rotations, device),
                    cov=_prep(cov3D_precomp, device), vm=_prep(viewmatrix, device), pm=_prep(projmatrix, device),
                    cam=_prep(campos, device), sh=_prep(sh, device), tanfov=_prep(tanfov, device), g=_prep(dL_dpix, device))
        M = int(keep["sh"].shape[-2]) if keep["sh"] is not None else 0
        use_sr = keep["cov"] is None
        # ONE allocation for every gradient, no fill from here: the library zeroes the four tensors it accumulates into (laid out
        # back to back: one fill) and writes every element of the others
        shapes = [("means2D", (V, P, 3)), ("conic", (V, P, 4)), ("colors", (S, P, 3) if keep["colors"] is not None else (V, P, 3)),
                  ("opacity", (S, P)), ("cov3D", (V, P, 6)), ("means3D", (S, P, 3)), ("sh", (S, P, max(M, 0), 3))]
        if use_sr:
            shapes += [("scales", (S, P, 3)), ("rotations", (S, P, 4))]
        sizes = [int(torch.Size(sh).numel()) for _, sh in shapes]
        flat = (torch.zeros if P == 0 else torch.empty)(sum(sizes), dtype=torch.float32, device=device)
        out, o = dict(scales=None, rotations=None), 0
        for (name, sh), n in zip(shapes, sizes):
            out[name] = flat[o:o + n].view(sh)
            o += n
        if P == 0:
            return out
        a = _native.DgsRasterBackwardArgs()
        a.P, a.D, a.M, a.width, a.height, a.V, a.views_per_set = P, int(degree), M, W, H, V, int(views_per_set)
        a.num_rendered = int(num_rendered)
        a.background = _ptr(keep["bg"]); a.means3D = _ptr(keep["means"]); a.shs = _ptr(keep["sh"])
        a.colors_precomp = _ptr(keep["colors"]); a.opacities = _ptr(keep["op"]); a.scales = _ptr(keep["scales"])
        a.rotations = _ptr(keep["rots"]); a.cov3D_precomp = _ptr(keep["cov"]); a.viewmatrix = _ptr(keep["vm"])
        a.projmatrix = _ptr(keep["pm"]); a.campos = _ptr(keep["cam"]); a.tanfov = _ptr(keep["tanfov"])
        a.tanfovx, a.tanfovy, a.scale_modifier = float(tanfovx), float(tanfovy), float(scale_modifier)
        a.debug, a.raw_activations = int(bool(debug)), int(bool(raw_activations))
        radii = radii.to(device=device, dtype=torch.int32).contiguous()
        a.radii = ctypes.c_void_p(radii.data_ptr()); a.dL_dpix = _ptr(keep["g"])
        a.geom_buffer, a.binning_buffer, a.img_buffer = _ptr(geom), _ptr(binning), _ptr(img)
        a.dL_dmeans2D, a.dL_dconic, a.dL_dcolors, a.dL_dcov3D = (_ptr(out[k]) for k in ("means2D", "conic", "colors", "cov3D"))
        a.dL_dopacity, a.dL_dmeans3D = _ptr(out["opacity"]), _ptr(out["means3D"])
        a.dL_dsh = _ptr(out["sh"]) if M else None
        a.dL_dscales, a.dL_drotations = (_ptr(out["scales"]), _ptr(out["rotations"])) if use_sr else (None, None)
        a.exact_exp = int(self.exact_exp)
        scratch = None
        if self.deterministic and int(num_rendered) > 0:
            nbytes = int(self.lib.dgs_raster_backward_scratch_bytes(P, W, H, V, int(num_rendered)))
            budget = self.deterministic_budget
            if device.type == "cuda":
                # ... and never more than 80 % of what the device can still give (free memory + what torch's allocator holds unused:
                # the previous step's scratch block comes back from there): a fuller or smaller GPU takes the atomic form instead of
                # failing with an out-of-memory error
                free_b, _ = torch.cuda.mem_get_info(device)
                budget = min(budget, int(0.8 * (free_b + torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device))))
            if nbytes <= budget:
                scratch = torch.empty(nbytes, dtype=torch.uint8, device=device)
                a.scratch, a.scratch_bytes = ctypes.c_void_p(scratch.data_ptr()), nbytes
        self.last_backward_deterministic = scratch is not None
        rc = self.lib.dgs_raster_backward(ctypes.byref(a), self._stream(device))
        self._check(rc)
        return out

    # -- Camera (gs_core.py:277-316), all views at once ------------------------------------
    def cameras_from_c2w(self, c2w, fxfycxcy, height, width, znear=0.01, zfar=100.0):
        """c2w [...,4,4], fxfycxcy [...,4] -> (viewmatrix [n,4,4], projmatrix [n,4,4], campos [n,3], tanfov [n,2])."""
        device = c2w.device
        c = _prep(c2w.reshape(-1, 4, 4), device)
        k = _prep(fxfycxcy.reshape(-1, 4), device)
        n = int(c.shape[0])
        view = torch.empty((n, 4, 4), dtype=torch.float32, device=device)
        proj = torch.empty((n, 4, 4), dtype=torch.float32, device=device)
        campos = torch.empty((n, 3), dtype=torch.float32, device=device)
        tanfov = torch.empty((n, 2), dtype=torch.float32, device=device)
        rc = self.lib.dgs_cameras_from_c2w(n, _ptr(c), _ptr(k), int(height), int(width), float(znear), float(zfar),
                                           _ptr(view), _ptr(proj), _ptr(campos), _ptr(tanfov), self._stream(device))
        self._check(rc)
        return view, proj, campos, tanfov

    def rays_from_c2w(self, c2w, fxfycxcy, height, width):
        """TransformInput (systems/utils.py:621-757): c2w [b,v,4,4], fxfycxcy [b,v,4] -> ray_o, ray_d [b,v,3,H,W]."""
        device = c2w.device
        lead = tuple(c2w.shape[:-2])
        c = _prep(c2w.reshape(-1, 4, 4), device)
        k = _prep(fxfycxcy.reshape(-1, 4), device)
        n = int(c.shape[0])
        ray_o = torch.empty((n, 3, int(height), int(width)), dtype=torch.float32, device=device)
        ray_d = torch.empty_like(ray_o)
        rc = self.lib.dgs_rays_from_c2w(n, _ptr(c), _ptr(k), int(height), int(width), _ptr(ray_o), _ptr(ray_d), self._stream(device))
        self._check(rc)
        shape = lead + (3, int(height), int(width))
        return ray_o.reshape(shape), ray_d.reshape(shape)

    def render_views(self, xyz, features, scaling, rotation, opacity, height, width, c2w, fxfycxcy, bg=None):
        """Forward-only batched render of RAW Gaussian parameters (what Renderer.forward / deferred_gaussian_render do
        per (sample, view) in the reference, renderer.py:34-92, gs_core.py:874-1016): xyz [B,P,3], features [B,P,M,3],
        scaling/rotation/opacity raw [B,P,3|4|1]; c2w [B,V,4,4]; fxfycxcy [B,V,4] -> [B,V,3,H,W] float32."""
        device = xyz.device
        B, V = int(c2w.shape[0]), int(c2w.shape[1])
        view, proj, campos, tanfov = self.cameras_from_c2w(c2w, fxfycxcy, height, width)
        if bg is None:
            bg = _white(device)                                        # render_opencv_cam default, gs_core.py:879
        M = int(features.shape[2])
        degree = int(round(M ** 0.5)) - 1
        out = self.forward_views(bg, xyz, None, opacity.reshape(B, -1), scaling, rotation, 1.0, None, view, proj, campos,
                                 tanfov, 0.0, 0.0, height, width, features, degree, False, False, views_per_set=V,
                                 raw_activations=True, planned=True, forward_only=True)
        return out[1].reshape(B, V, 3, int(height), int(width))

    # -- _C.mark_visible ------------------------------------------------------------------
    def mark_visible(self, means3D, viewmatrix, projmatrix):
        device = means3D.device
        P = int(means3D.shape[0])
        present = torch.zeros((P,), dtype=torch.bool, device=device)
        if P:
            m, vm, pm = _prep(means3D, device), _prep(viewmatrix, device), _prep(projmatrix, device)
            rc = self.lib.dgs_mark_visible(P, _ptr(m), _ptr(vm), _ptr(pm), ctypes.c_void_p(present.data_ptr()),
                                           self._stream(device))
            self._check(rc)
        return present

    # -- parity-test introspection -----------------------------------------------------------
    def state_read(self, name, P, W, H, V, num_rendered, geom, binning, img, dtype, count):
        device = geom.device
        dst = torch.empty((count,), dtype=dtype, device=device)
        n = self.lib.dgs_raster_state_read(name.encode(), P, W, H, V, int(num_rendered), _ptr(geom), _ptr(binning), _ptr(img),
                                           ctypes.c_void_p(dst.data_ptr()), dst.numel() * dst.element_size(),
                                           self._stream(device))
        if n < 0:
            self._check(int(n))
        return dst[: int(n) // dst.element_size()]


class _RenderViews(torch.autograd.Function):
    """Differentiable batched render of raw Gaussian parameters: what DeferredGaussianRender (gs_core.py:949-1064) does with
    b*v per-view calls and a recompute in backward, as ONE forward and ONE backward launch sequence.  The forward state
    buffers are kept for backward (no second forward pass: MI355X has the HBM for it)."""

    @staticmethod
    def forward(ctx, backend, xyz, features, scaling, rotation, opacity, c2w, fxfycxcy, height, width, bg):
        device = xyz.device
        B, V = int(c2w.shape[0]), int(c2w.shape[1])
        view, proj, campos, tanfov = backend.cameras_from_c2w(c2w, fxfycxcy, height, width)
        M = int(features.shape[2])
        degree = int(round(M ** 0.5)) - 1
        f = lambda t: t.detach().to(torch.float32).contiguous()
        xyz_, sh_, sc_, ro_, op_ = f(xyz), f(features), f(scaling), f(rotation), f(opacity).reshape(B, -1)
        n, color, radii, geom, binning, img = backend.forward_views(
            bg, xyz_, None, op_, sc_, ro_, 1.0, None, view, proj, campos, tanfov, 0.0, 0.0, height, width, sh_, degree,
            False, False, views_per_set=V, raw_activations=True, planned=True)
        ctx.backend, ctx.meta = backend, (B, V, int(height), int(width), degree, n)
        ctx.save_for_backward(bg, xyz_, sh_, sc_, ro_, op_, view, proj, campos, tanfov, radii, geom, binning, img)
        return color.reshape(B, V, 3, int(height), int(width))

    @staticmethod
    def backward(ctx, grad):
        bg, xyz_, sh_, sc_, ro_, op_, view, proj, campos, tanfov, radii, geom, binning, img = ctx.saved_tensors
        B, V, H, W, degree, n = ctx.meta
        g = ctx.backend.backward_views(bg, xyz_, radii, None, op_, sc_, ro_, 1.0, None, view, proj, campos, tanfov, 0.0, 0.0,
                                       grad.reshape(B * V, 3, H, W), sh_, degree, geom, n, binning, img, False,
                                       views_per_set=V, raw_activations=True)
        return (None, g["means3D"], g["sh"], g["scales"], g["rotations"], g["opacity"].reshape(B, -1, 1), None, None, None,
                None, None)


def render_views_autograd(backend, xyz, features, scaling, rotation, opacity, height, width, c2w, fxfycxcy, bg=None):
    """[B,P,..] raw Gaussian parameters -> [B,V,3,H,W]; differentiable w.r.t. the five parameter tensors."""
    if bg is None:
        bg = _white(xyz.device)
    return _RenderViews.apply(backend, xyz, features, scaling, rotation, opacity, c2w.float(), fxfycxcy.float(), height, width, bg)


_default = None
_WHITE = {}


def _white(device):
    """The default background, one tensor per device (a torch.ones(3) per render call was a fill launch of its own in every step)."""
    key = (device.type, device.index)
    if key not in _WHITE:
        _WHITE[key] = torch.ones(3, dtype=torch.float32, device=device)
    return _WHITE[key]


def default_backend():
    global _default
    if _default is None:
        _default = RasterBackend()
    return _default
