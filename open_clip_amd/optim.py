"""AdamW on the HIP path: the optimizer step of ``train.py:181-182`` -- gradient clipping,
AdamW and the refresh of the bf16 GEMM-operand copies of the weights -- as ONE launch over the whole model.

``NativeAdamW`` is a ``torch.optim.Optimizer`` with torch.optim.AdamW's semantics (decoupled weight decay, bias
correction; verified against it in tests/test_kernels_gpu.py).  Every step it describes the parameters to the device
as a table of 88-byte records (``ocn_adamw_multi``, include/openclip_hip.h) -- gradient pointers and per-group
learning rates change from step to step, the chunk list that cuts the tensors into 8192-element ranges / 64x64 tiles
does not -- and launches one kernel.  When the model's ``_WeightCache`` objects are passed (``weight_caches=``), the
kernel also rewrites the cached bf16 copies ([out,in] and the transposed [in,out]) in the same pass, so no cast
kernels run in the next forward.  ``grad_clip_norm`` fuses ``torch.nn.utils.clip_grad_norm_`` (train.py:181): one
multi-tensor sum of squares, the coefficient is applied inside the AdamW kernel (``.grad`` is left unscaled).
``param_groups_like_reference`` reproduces the reference's grouping rule (optim.py:67-77,178-208: 1-D params and
``model.no_weight_decay()`` names get weight_decay 0)."""
import numpy as np
import torch

from . import _lib, ops

_ENTRY = np.dtype([("w", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("w16n", "<u8"), ("w16t", "<u8"), ("numel", "<i8"),
                   ("rows", "<i4"), ("cols", "<i4"), ("mode", "<i4"), ("pad", "<i4"),
                   ("lr", "<f4"), ("wd", "<f4"), ("bc1", "<f4"), ("bc2_sqrt", "<f4")])
assert _ENTRY.itemsize == 88
_CHUNK = 8192


def param_groups_like_reference(model, weight_decay=0.2):
    skip = set(model.no_weight_decay()) if hasattr(model, "no_weight_decay") else set()
    decay, no_decay = [], []
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        n = n[len("module."):] if n.startswith("module.") else n
        (no_decay if (p.ndim <= 1 or n in skip) else decay).append(p)  # optim.py:67-77 exclude_from_wd + no_weight_decay()
    return [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": weight_decay}]


def weight_caches_of(model):
    """the ``_WeightCache`` objects of a NativeCLIP (both towers), for ``NativeAdamW(weight_caches=...)``"""
    m = model.module if hasattr(model, "module") else model
    return [c for c in (getattr(m, "_cache", None), getattr(getattr(m, "visual", None), "_cache", None)) if c is not None]


class NativeAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=5e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.2, grad_clip_norm=None, weight_caches=(),
                 fused=True, deterministic=False):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.grad_clip_norm = grad_clip_norm
        # the squared gradient norm of ``grad_clip_norm`` summed in a fixed order (per-chunk partials folded by one workgroup) instead of with
        # fp32 atomics: the optimizer's share of NativeCLIP(deterministic=True) -- the clipped update is then bit-reproducible too
        self.deterministic = bool(deterministic)
        self.weight_caches = list(weight_caches)
        self.fused = fused
        self._plan_key = None
        self._plan = None
        self.last_grad_norm_sq = None  # device scalar of the most recent step when grad_clip_norm is set

    # ---- per-tensor reference path (one ocn_adamw_step launch per tensor) ------------------------------------------
    @torch.no_grad()
    def _step_per_tensor(self):
        coef = None
        if self.grad_clip_norm is not None:
            acc = None
            for group in self.param_groups:
                for p in group["params"]:
                    if p.grad is not None:
                        acc = torch.zeros(1, dtype=torch.float32, device=p.device) if acc is None else acc
                        ops.sumsq_accum(p.grad.contiguous().view(-1), acc)
            if acc is not None:
                self.last_grad_norm_sq = acc
                coef = torch.clamp(self.grad_clip_norm / (acc.sqrt() + 1e-6), max=1.0)
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self._state_of(p)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                ops.adamw_step(p.view(-1), g.view(-1), st["exp_avg"].view(-1), st["exp_avg_sq"].view(-1), group["lr"], b1, b2,
                               group["eps"], group["weight_decay"], st["step"], clip_coef=coef)
                torch.autograd.graph.increment_version(p)  # raw-pointer update: let the bf16 weight cache see it

    def _state_of(self, p):
        st = self.state[p]
        if not st:
            st["step"] = 0
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        return st

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._plan = self._plan_key = None  # the moments are new tensors: rebuild the device table on the next step
        for st in self.state.values():      # torch stores `step` as a tensor in its own AdamW checkpoints: accept both
            if torch.is_tensor(st.get("step")):
                st["step"] = int(st["step"].item())

    # ---- fused path ---------------------------------------------------------------------------------------------------
    def _copies_of(self, p):
        n = t = None
        for c in self.weight_caches:
            n = c.peek(p, "n") if n is None else n
            t = c.peek(p, "t") if t is None else t
        return n, t

    def _build_plan(self, active):
        """static part of the step: state tensors, operand copies, modes, chunk list (rebuilt when the set of tensors with
        a gradient, or the set of cached operand copies, changes)"""
        dev = active[0][1].device
        ent = np.zeros(len(active), dtype=_ENTRY)
        chunks, copies = [], []
        for i, (group, p) in enumerate(active):
            st = self._state_of(p)
            n16, t16 = (None, None) if p.ndim != 2 and not (p.ndim == 4) else self._copies_of(p)
            rows = p.shape[0] if p.ndim >= 2 else 1
            cols = p.numel() // rows
            e = ent[i]
            e["w"], e["m"], e["v"] = p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            e["w16n"] = 0 if n16 is None else n16.data_ptr()
            e["w16t"] = 0 if t16 is None else t16.data_ptr()
            e["numel"], e["rows"], e["cols"] = p.numel(), rows, cols
            tile = t16 is not None and rows % 64 == 0 and cols % 64 == 0
            if t16 is not None and not tile:  # odd-shaped weight: keep its transposed copy on the cast path
                e["w16t"] = 0
                t16 = None
            e["mode"] = 1 if tile else (0 if p.data_ptr() % 16 == 0 else 2)
            k = (rows // 64) * (cols // 64) if tile else -(-p.numel() // _CHUNK)
            c = np.empty((k, 2), dtype=np.int32)
            c[:, 0], c[:, 1] = i, np.arange(k, dtype=np.int32)
            chunks.append(c)
            copies.append((n16, t16))
        chunks = np.concatenate(chunks, axis=0)
        # the record table travels host -> device every step through one of TWO pinned buffers used alternately: the buffer a step fills
        # was last read by the upload of two steps ago, so the wait below never blocks a host that runs a step ahead of the device
        plan = {"entries": ent, "n_chunks": int(chunks.shape[0]), "copies": copies, "copied": [None, None], "turn": 0,
                "chunks_dev": torch.from_numpy(chunks).to(dev),
                "entries_host": [torch.empty(len(active) * _ENTRY.itemsize, dtype=torch.uint8).pin_memory() for _ in range(2)],
                "entries_dev": torch.empty(len(active) * _ENTRY.itemsize, dtype=torch.uint8, device=dev)}
        return plan

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        if not self.fused:
            self._step_per_tensor()
            return loss
        active = [(g, p) for g in self.param_groups for p in g["params"] if p.grad is not None]
        if not active:
            return loss
        key = tuple((id(p), tuple(p.shape)) + (tuple(map(id, self._copies_of(p))) if p.ndim in (2, 4) else ()) for _, p in active)
        if key != self._plan_key:
            self._plan, self._plan_key = self._build_plan(active), key
        plan = self._plan
        turn = plan["turn"]
        plan["turn"] = 1 - turn
        if plan["copied"][turn] is not None and not plan["copied"][turn].query():
            plan["copied"][turn].synchronize()  # the upload of two steps ago out of this pinned buffer (in practice long finished)
        ent = plan["entries"]
        b1, b2 = active[0][0]["betas"]
        eps = active[0][0]["eps"]
        grads = []
        for i, (group, p) in enumerate(active):
            if group["betas"] != (b1, b2) or group["eps"] != eps:
                raise RuntimeError("NativeAdamW (fused): all param groups must share betas and eps")
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            grads.append(g)
            st = self._state_of(p)
            st["step"] += 1
            e = ent[i]
            # every pointer of the record is re-read every step: load_state_dict(), model.to(), `p.data = ...` replace the
            # parameter / moment tensors behind an unchanged id(p), and a cached address would then update freed memory
            m, v = st["exp_avg"], st["exp_avg_sq"]
            if not (m.is_contiguous() and v.is_contiguous() and p.is_contiguous()):
                raise RuntimeError("NativeAdamW (fused): parameters and their moments must be contiguous")
            if m.device != p.device or m.dtype != torch.float32 or v.dtype != torch.float32:
                st["exp_avg"], st["exp_avg_sq"] = m, v = m.to(p.device, torch.float32), v.to(p.device, torch.float32)
            e["w"], e["m"], e["v"], e["g"] = p.data_ptr(), m.data_ptr(), v.data_ptr(), g.data_ptr()
            if e["mode"] != 1:
                e["mode"] = 0 if (p.data_ptr() % 16 == 0 and g.data_ptr() % 16 == 0 and m.data_ptr() % 16 == 0 and v.data_ptr() % 16 == 0) else 2
            e["lr"], e["wd"] = group["lr"], group["weight_decay"]
            e["bc1"] = 1.0 - b1 ** st["step"]
            e["bc2_sqrt"] = (1.0 - b2 ** st["step"]) ** 0.5
        host = plan["entries_host"][turn]
        host.numpy()[:] = ent.view(np.uint8).reshape(-1)
        plan["entries_dev"].copy_(host, non_blocking=True)
        plan["copied"][turn] = torch.cuda.Event()
        plan["copied"][turn].record()
        stream = torch.cuda.current_stream().cuda_stream
        gn, max_norm = 0, 0.0
        if self.grad_clip_norm is not None:
            acc = torch.zeros(1, dtype=torch.float32, device=plan["entries_dev"].device)
            ws = torch.empty(plan["n_chunks"], dtype=torch.float32, device=acc.device) if self.deterministic else None
            _lib.call("ocn_sumsq_multi", plan["entries_dev"].data_ptr(), plan["chunks_dev"].data_ptr(), plan["n_chunks"], acc.data_ptr(),
                      0 if ws is None else ws.data_ptr(), stream)
            self.last_grad_norm_sq = acc
            gn, max_norm = acc.data_ptr(), float(self.grad_clip_norm)
        _lib.call("ocn_adamw_multi", plan["entries_dev"].data_ptr(), plan["chunks_dev"].data_ptr(), plan["n_chunks"], float(b1), float(b2),
                  float(eps), gn, max_norm, stream)
        params = [p for _, p in active]
        torch.autograd.graph.increment_version(params)  # raw-pointer update
        for p, (n16, t16) in zip(params, plan["copies"]):  # the operand copies were rewritten in the same launch
            for c in self.weight_caches:
                c.mark_fresh(p, n16, t16)
        return loss
