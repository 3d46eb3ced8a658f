"""ViTEngine — frozen integer DeiT/ViT forward on one MI355X through the C-ABI.

`forward` / `capture` hand the whole batch to the native runner (`ivit_vit_forward`,
csrc/ivit_model.h): one C call per batch, slices on internal HIP streams, optional hipGraph.
`forward_ops` issues the same kernels one C-ABI call at a time from Python (used by the
per-operator timing in bench.py and by the parity tests of the unfused attention path).
Both follow the call order of the reference `VisionTransformer.forward`
(models/vit_quant.py:254-282; Block :130-143; Attention :59-88; Mlp
layers_quant.py:144-153) with the QuantLinear->QuantAct, IntLayerNorm->QuantAct,
IntGELU->QuantAct and QuantAct->QuantAct(identity) pairs fused into single kernels.
torch is used for device memory and streams only.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .freeze import freeze_vit

_P = ctypes.c_void_p


def _align(n, a=256):
    return (n + a - 1) // a * a


def pack_constants(consts):
    """One contiguous byte blob + table {name: (offset, dtype str, shape)} — the unit
    that is broadcast over RCCL to the other ranks."""
    table, off = {}, 0
    for k in sorted(consts):
        a = np.ascontiguousarray(consts[k])
        table[k] = (off, a.dtype.str, a.shape)
        off = _align(off + a.nbytes)
    blob = np.zeros(off, np.uint8)
    for k, (o, _, _) in table.items():
        a = np.ascontiguousarray(consts[k])
        blob[o:o + a.nbytes] = a.view(np.uint8).reshape(-1)
    return blob, table


def _dy(arr):
    """host [1,2] float64 -> by-value struct ivit_dyadic"""
    return _lib.Dyadic(float(arr[0, 0]), float(arr[0, 1]))


def host_scalars(blob, table):
    """the by-value entries of a packed blob (host numpy): scalar dyadics and the Shiftmax table metadata"""
    host = {}
    for k, (o, dt, shp) in table.items():
        if dt == "<f8" and shp == (1, 2):
            host[k] = blob[o:o + 16].view(np.float64).reshape(1, 2).copy()
        elif k.endswith("exp_meta"):
            host[k] = blob[o:o + 12].view(np.int32).copy()
    return host


def vit_native_params(cfg, table, f32, host, base):
    """(ivit_vit_config, ivit_vit_params, keep-alive) for ivit_vit_create — or for its CPU twin: `base` is the address of
    the packed constants blob, in device memory for the library, in host memory for oracle/ivit_twin.c"""
    L = _lib
    ptr = lambda name: base + table[name][0]
    dy = lambda name: _dy(host[name])
    blocks = (L.VitBlock * cfg.depth)()
    for i in range(cfg.depth):
        p, b = f"blocks.{i}.", blocks[i]
        b.s_ln1, b.n1_bias_int, b.n1_sc, b.n1_dy = f32[p + "ln1.s"], ptr(p + "norm1.bias_int"), ptr(p + "norm1.sc"), ptr(p + "norm1.dy")
        b.qkv_w, b.qkv_b, b.qkv_dy = ptr(p + "attn.qkv.w"), ptr(p + "attn.qkv.b"), ptr(p + "attn.qkv.dy")
        b.dy_qk, b.s_softmax, b.dy_pv = dy(p + "attn.dy_qk"), f32[p + "attn.s_softmax"], dy(p + "attn.dy_pv")
        if p + "attn.exp_meta" in host:        # Shiftmax tables for this layer's scale
            meta = host[p + "attn.exp_meta"]
            b.exp_aq, b.exp_t, b.exp_cls = ptr(p + "attn.exp_aq"), ptr(p + "attn.exp_t"), ptr(p + "attn.exp_cls")
            b.exp_nc, b.exp_tcount, b.exp_dmin = int(meta[0]), int(meta[1]), int(meta[2])
        b.proj_w, b.proj_b, b.proj_dy = ptr(p + "attn.proj.w"), ptr(p + "attn.proj.b"), ptr(p + "attn.proj.dy")
        b.res1_main, b.res1_res = dy(p + "res1.dy_main"), dy(p + "res1.dy_res")
        b.s_ln2, b.n2_bias_int, b.n2_sc, b.n2_dy = f32[p + "ln2.s"], ptr(p + "norm2.bias_int"), ptr(p + "norm2.sc"), ptr(p + "norm2.dy")
        b.fc1_w, b.fc1_b, b.fc1_dy = ptr(p + "mlp.fc1.w"), ptr(p + "mlp.fc1.b"), ptr(p + "mlp.fc1.dy")
        b.s_gelu, b.dy_gelu = f32[p + "mlp.s_gelu"], dy(p + "mlp.dy_gelu")
        b.fc2_w, b.fc2_b, b.fc2_dy = ptr(p + "mlp.fc2.w"), ptr(p + "mlp.fc2.b"), ptr(p + "mlp.fc2.dy")
        b.res2_main, b.res2_res = dy(p + "res2.dy_main"), dy(p + "res2.dy_res")
    prm = L.VitParams()
    prm.pe_w, prm.pe_b, prm.pe_dy = ptr("patch_embed.proj.w"), ptr("patch_embed.proj.b"), ptr("patch_embed.proj.dy")
    prm.z_cls, prm.pos, prm.dy_x, prm.dy_pos = ptr("z_cls"), ptr("pos"), dy("embed.dy_x"), dy("embed.dy_pos")
    prm.blocks_host = ctypes.cast(blocks, ctypes.POINTER(L.VitBlock))
    prm.s_ln, prm.n_bias_int, prm.n_sc, prm.n_dy = f32["ln.s"], ptr("norm.bias_int"), ptr("norm.sc"), ptr("norm.dy")
    prm.head_w, prm.head_b = ptr("head.w"), ptr("head.b")
    c = L.VitConfig(cfg.img_size, cfg.patch_size, cfg.in_chans, cfg.embed_dim, cfg.depth, cfg.num_heads,
                    cfg.hidden_dim, cfg.num_classes)
    return c, prm, blocks


class ViTEngine:
    def __init__(self, cfg, consts, f32, device="cuda:0", blob=None, table=None):
        """consts/f32 from freeze.freeze_vit (rank 0) — or a pre-packed (blob, table)
        received from a broadcast (then `consts` may be None)."""
        self.cfg = cfg
        self.device = torch.device(device)
        if not torch.cuda.is_available():
            raise _lib.IvitError("ViTEngine needs a HIP device; the product path has no CPU fallback")
        torch.cuda.set_device(self.device)
        if blob is None:
            blob, table = pack_constants(consts)
        self.table = table
        self.f32 = {k: float(np.float32(v)) for k, v in f32.items()}
        if isinstance(blob, np.ndarray):
            self.blob = torch.from_numpy(blob).to(self.device)
        else:
            self.blob = blob  # already a device uint8 tensor
        # scalar dyadics are passed by value: keep host copies
        self.host = host_scalars(blob if isinstance(blob, np.ndarray) else self.blob.cpu().numpy(), table)
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.h = _lib.Handle(dev_index, torch.cuda.current_stream(self.device).cuda_stream)
        self._ws = {}
        self.fused_attention = (cfg.head_dim == 64 and cfg.num_tokens <= 640)
        self.use_exp_tables = True      # forward_ops only: False issues the arithmetic Shiftmax (cross-check)
        # per-layer ShiftGELU(+requant) tables, built on-device by the faithful kernel code
        self.gelu_tab = torch.empty(cfg.depth, 65536, dtype=torch.int8, device=self.device)
        for i in range(cfg.depth):
            p = f"blocks.{i}."
            self.h.call("ivit_shiftgelu_build_table", self.f32[p + "mlp.s_gelu"], _dy(self.host[p + "mlp.dy_gelu"]),
                        _P(self.gelu_tab[i].data_ptr()))

        # Shiftmax row tables (one gather per score, ivit_attention_fused_rowlut) for the layers whose table lines fit 64 entries
        # and whose requant multipliers are in the kernel's fast range — the per-operator path makes the runner's choice
        self.use_row_tables = True
        self.v_row_major = True         # forward_ops: v row-major into the row-table attention (False: v^T as before)
        self.fuse_ln_qkv = True         # forward_ops: norm1 in the qkv GEMM's prologue where the plan is prepared (False: two launches)
        self.fuse_ln_mlp = True         # forward_ops: norm2 in the head of the fused Mlp's launch (False: two launches)
        self.fuse_patch_embed = True    # forward_ops: ivit_patch_embed (False: im2col, GEMM, embed_finish)
        self._qkv_prepared = set()
        self.rowtab = {}
        if self.fused_attention:
            for i in range(cfg.depth):
                p = f"blocks.{i}."
                if p + "attn.exp_meta" not in self.host:
                    continue
                meta = self.host[p + "attn.exp_meta"]
                dqk, dpv = self.host[p + "attn.dy_qk"], self.host[p + "attn.dy_pv"]
                if 1 - int(meta[2]) <= 64 and abs(dqk[0, 0] * dqk[0, 1]) < 512.0 and abs(dpv[0, 0] * dpv[0, 1]) < 512.0:
                    t = torch.empty(256, 64, dtype=torch.float32, device=self.device)
                    self.h.call("ivit_shiftmax_rowtable", self.ptr(p + "attn.exp_aq"), self.ptr(p + "attn.exp_t"), self.ptr(p + "attn.exp_cls"),
                                int(meta[0]), int(meta[1]), int(meta[2]), _P(t.data_ptr()))
                    self.rowtab[i] = t

        # frozen-QuantLinear plans of the per-operator path (forward_ops / plan()): built on first use by build_op_plans()
        # — plan creation allocates and synchronises, so callers that time or capture forward_ops call it beforehand; the
        # native runner (forward) holds its own plans, and an engine that only runs forward() no longer keeps a second
        # copy of the fragment-ordered Mlp weights (14 MB for DeiT-S).  use_plans = False issues the unplanned kernels
        self.use_plans = True
        self.use_fused_mlp = True       # forward_ops: ivit_mlp_fused_planned where a fused plan exists (D = 384)
        self._plans = {}
        self._mlp_plans = {}
        self._build_native()

    MAX_SLICES = 8

    def build_op_plans(self):
        """ivit_linear_plan_create / ivit_mlp_plan_create for every block (idempotent)."""
        if self._plans:
            return
        cfg = self.cfg
        D, Hd = cfg.embed_dim, cfg.hidden_dim
        for i in range(cfg.depth):
            p = f"blocks.{i}."
            for name, N, K in ((p + "attn.qkv", 3 * D, D), (p + "attn.proj", D, D), (p + "mlp.fc1", Hd, D), (p + "mlp.fc2", D, Hd)):
                self._plans[name] = self.h.linear_plan(self.ptr(name + ".w"), self.ptr(name + ".b"), self.ptr(name + ".dy"), N, K)
                if name.endswith("attn.qkv") and D == 384 and D // cfg.num_heads == 64 and \
                        self.h.lib.ivit_linear_plan_prepare_ws(self.h.h, self._plans[name].p) == 0:
                    self._qkv_prepared.add(name)       # gemm_ws_qkv_kernel's weight order: ldv = 0 calls and the fused norm1 form use it
                if name.endswith("attn.proj") and D == 384:     # ... and attn.proj + the residual QuantAct (the runner's choice)
                    self.h.lib.ivit_linear_plan_prepare_ws(self.h.h, self._plans[name].p)
            mp = _P()
            if self.h.lib.ivit_mlp_plan_create(self.h.h, self._plans[p + "mlp.fc1"].p, self._plans[p + "mlp.fc2"].p, ctypes.byref(mp)) == 0:
                self._mlp_plans[i] = mp

    def plan(self, prefix):
        """frozen-QuantLinear plan (ivit_linear_plan_create) of the layer `prefix`; the native runner holds its own."""
        self.build_op_plans()
        return self._plans[prefix].p

    def _build_native(self):
        """ivit_vit_create: hand the runner device pointers into the blob + host scalars."""
        c, prm, self._native_keep = vit_native_params(self.cfg, self.table, self.f32, self.host, self.blob.data_ptr())
        self.model = _P()
        self.h._check(self.h.lib.ivit_vit_create(self.h.h, ctypes.byref(c), ctypes.byref(prm), self.MAX_SLICES,
                                                 ctypes.byref(self.model)), "ivit_vit_create")
        self._native_ws = {}

    def __del__(self):
        try:
            if getattr(self, "model", None):
                self.h.lib.ivit_vit_destroy(self.model)
                self.model = None
            for mp in getattr(self, "_mlp_plans", {}).values():
                self.h.lib.ivit_mlp_plan_destroy(mp)
            self._mlp_plans = {}
            for pl in getattr(self, "_plans", {}).values():
                pl.close()
        except Exception:
            pass

    def _native_buffers(self, B, nslices):
        key = (B, nslices)
        if key not in self._native_ws:
            # bounded: at most 4 (batch, slices) shapes that no captured graph refers to stay resident.  A graph has its
            # workspace pointer baked in: those entries are pinned (self._graph_keys) and never evicted
            free = [k for k in self._native_ws if k not in getattr(self, "_graph_keys", set())]
            if len(free) >= 4:
                self._native_ws.pop(free[0])
        if key not in self._native_ws:
            n = ctypes.c_size_t()
            self.h._check(self.h.lib.ivit_vit_workspace_bytes(self.model, B, nslices, ctypes.byref(n)), "ivit_vit_workspace_bytes")
            ws = torch.empty(n.value, dtype=torch.uint8, device=self.device)
            logits = torch.empty(B, self.cfg.num_classes, dtype=torch.int32, device=self.device)
            self.h._check(self.h.lib.ivit_vit_workspace_init(self.model, _P(ws.data_ptr()), n.value, B, nslices), "ivit_vit_workspace_init")
            self._native_ws[key] = (ws, logits)
        return self._native_ws[key]

    def forward(self, images, nslices=1, copy=False):
        """images: int8 device tensor [B, C, H, W] (already quantised, scale s_in) -> int32 logits
        [B, num_classes] (head accumulators).  One native call; nslices > 1 cuts the batch into slices
        on the runner's internal HIP streams (VALU-bound kernels of one slice share the chip with the
        MFMA-bound GEMMs of another).  Same integers for every nslices.

        The returned tensor is the engine's OWN output buffer for this (batch, nslices): the next forward /
        graph replay of the same shape overwrites it (nothing is allocated per call).  Pass copy=True — or
        clone it — when results of several batches are kept (an eval loop collecting logits)."""
        assert images.dtype == torch.int8 and images.is_contiguous() and images.device == self.device
        self.h.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        B = images.shape[0]
        nslices = max(1, min(int(nslices), B, self.MAX_SLICES))
        ws, logits = self._native_buffers(B, nslices)
        self.h._check(self.h.lib.ivit_vit_forward(self.model, _P(images.data_ptr()), B, nslices, _P(ws.data_ptr()),
                                                  ws.numel(), _P(logits.data_ptr())), "ivit_vit_forward")
        return logits.clone() if copy else logits

    def forward_streams(self, images, nstreams=2):
        return self.forward(images, nslices=nstreams)

    def capture(self, images, nstreams=1):
        """hipGraph of one forward on fixed buffers (ivit_vit_graph_create).  Returns a callable that
        replays it and returns the logits tensor."""
        B = images.shape[0]
        nslices = max(1, min(int(nstreams), B, self.MAX_SLICES))
        ws, logits = self._native_buffers(B, nslices)
        if not hasattr(self, "_gstream"):
            self._gstream = torch.cuda.Stream(self.device)
        torch.cuda.synchronize(self.device)
        self.h.set_stream(self._gstream.cuda_stream)
        g = _P()
        self.h._check(self.h.lib.ivit_vit_graph_create(self.model, _P(images.data_ptr()), B, nslices, _P(ws.data_ptr()),
                                                       ws.numel(), _P(logits.data_ptr()), ctypes.byref(g)), "ivit_vit_graph_create")
        # the graph replays on `ws` / `logits` / `images`: all three live as long as the replay closure does, and the
        # workspace entry is pinned against eviction
        self._graphs = getattr(self, "_graphs", []) + [(g, ws, logits, images)]
        self._graph_keys = getattr(self, "_graph_keys", set()) | {(B, nslices)}
        lib, gs, dev = self.h.lib, self._gstream, self.device

        def replay(_keep=(ws, logits, images)):
            cur = torch.cuda.current_stream(dev)
            gs.wait_stream(cur)
            self.h.set_stream(gs.cuda_stream)
            self.h._check(lib.ivit_graph_launch(g), "ivit_graph_launch")
            cur.wait_stream(gs)
            return logits
        return replay

    @classmethod
    def from_float(cls, cfg, weights, scales, device="cuda:0"):
        consts, f32 = freeze_vit(cfg, weights, scales)
        return cls(cfg, consts, f32, device)

    def ptr(self, name):
        return _P(self.blob.data_ptr() + self.table[name][0])

    def head_scale(self):
        o, dt, shp = self.table["head.scale"]
        return self.blob[o:o + 4 * shp[0]].cpu().numpy().view(np.float32).copy()

    def workspace(self, B, key=None):
        wkey = (B, key)
        if wkey in self._ws:
            return self._ws[wkey]
        cfg, dev = self.cfg, self.device
        T, D, H, dh, Hd = cfg.num_tokens, cfg.embed_dim, cfg.num_heads, cfg.head_dim, cfg.hidden_dim
        ld = (T + 15) // 16 * 16
        M = B * T
        Kp = cfg.in_chans * cfg.patch_size ** 2
        e = lambda *s, dt: torch.empty(s, dtype=dt, device=dev)
        ws = dict(
            ld=ld,
            patches=e(B * cfg.num_patches, Kp, dt=torch.int8),
            patch16=e(B * cfg.num_patches, D, dt=torch.int16),
            xa=e(M, D, dt=torch.int16), xb=e(M, D, dt=torch.int16),
            a8=e(M, D, dt=torch.int8),
            q=e(B * H, T, dh, dt=torch.int8), k=e(B * H, T, dh, dt=torch.int8),
            vt=torch.zeros(B * H, dh, ld, dtype=torch.int8, device=dev),
            s8=e(B * H, T, ld, dt=torch.int8),
            p16=torch.zeros(B * H, T, ld, dtype=torch.int16, device=dev),  # uint16 payload
            ctx8=e(M, D, dt=torch.int8),
            h8=e(M, Hd, dt=torch.int8), g8=e(M, Hd, dt=torch.int8),
            cls8=e(B, D, dt=torch.int8),
            logits=e(B, cfg.num_classes, dt=torch.int32),
        )
        self._ws[wkey] = ws
        return ws

    def forward_ops(self, images, ws_key=None):
        """The same forward issued one C-ABI call per operator from Python (per-operator timing,
        unfused-attention parity).  Returns int32 logits [B, num_classes]."""
        cfg, call, f32, hc = self.cfg, self.h.call, self.f32, self.host
        assert images.dtype == torch.int8 and images.is_contiguous() and images.device == self.device
        self.h.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        if self.use_plans:
            self.build_op_plans()
        B = images.shape[0]
        T, D, H, dh, Hd = cfg.num_tokens, cfg.embed_dim, cfg.num_heads, cfg.head_dim, cfg.hidden_dim
        M = B * T
        ws = self.workspace(B, ws_key)
        ld = ws["ld"]
        P = lambda t: _P(t.data_ptr())
        Kp = cfg.in_chans * cfg.patch_size ** 2
        x, y = ws["xa"], ws["xb"]
        # the runner's choice: the whole PatchEmbed front end as one GEMM launch where it applies (16 x 16 patches, fast multipliers)
        one = self.fuse_patch_embed and self.h.try_call(
            "ivit_patch_embed", P(images), B, cfg.in_chans, cfg.img_size, cfg.img_size, cfg.patch_size, self.ptr("patch_embed.proj.w"),
            self.ptr("patch_embed.proj.b"), self.ptr("patch_embed.proj.dy"), self.ptr("z_cls"), self.ptr("pos"), _dy(hc["embed.dy_x"]),
            _dy(hc["embed.dy_pos"]), P(x), D)
        if not one:
            call("ivit_im2col_patch", P(images), B, cfg.in_chans, cfg.img_size, cfg.img_size, cfg.patch_size,
                 P(ws["patches"]))
            call("ivit_linear_i8_requant", P(ws["patches"]), self.ptr("patch_embed.proj.w"),
                 self.ptr("patch_embed.proj.b"), self.ptr("patch_embed.proj.dy"), 16, P(ws["patch16"]),
                 B * cfg.num_patches, D, Kp)
            call("ivit_embed_finish", P(ws["patch16"]), self.ptr("z_cls"), self.ptr("pos"), _dy(hc["embed.dy_x"]),
                 _dy(hc["embed.dy_pos"]), P(x), B, T, D)
        for i in range(cfg.depth):
            p = f"blocks.{i}."
            # the runner's choice (csrc/ivit_model.h): v ROW-major (ldv = 0) between the planned qkv GEMM and the row-table attention
            row_attn = self.fused_attention and i in self.rowtab and self.use_exp_tables and self.use_row_tables
            ldv = 0 if (row_attn and self.use_plans and self.v_row_major) else ld
            # ... and norm1 inside that GEMM's prologue where its plan is prepared (D = 384, dh = 64: ivit_layernorm_linear_i8_qkv_planned)
            fused_ln = ldv == 0 and self.fuse_ln_qkv and (p + "attn.qkv") in self._qkv_prepared
            if not fused_ln:
                call("ivit_layernorm_requant", P(x), M, D, D, f32[p + "ln1.s"], self.ptr(p + "norm1.bias_int"),
                     self.ptr(p + "norm1.sc"), self.ptr(p + "norm1.dy"), P(ws["a8"]))
            if fused_ln:
                call("ivit_layernorm_linear_i8_qkv_planned", self.plan(p + "attn.qkv"), P(x), f32[p + "ln1.s"], self.ptr(p + "norm1.bias_int"),
                     self.ptr(p + "norm1.sc"), self.ptr(p + "norm1.dy"), P(ws["q"]), P(ws["k"]), P(ws["vt"]), B, T, H, dh)
            elif self.use_plans:
                call("ivit_linear_i8_qkv_planned", self.plan(p + "attn.qkv"), P(ws["a8"]), P(ws["q"]), P(ws["k"]),
                     P(ws["vt"]), B, T, H, dh, ldv)
            else:
                call("ivit_linear_i8_qkv", P(ws["a8"]), self.ptr(p + "attn.qkv.w"), self.ptr(p + "attn.qkv.b"),
                     self.ptr(p + "attn.qkv.dy"), P(ws["q"]), P(ws["k"]), P(ws["vt"]), B, T, H, dh, ld)
            if self.fused_attention:
                if row_attn:
                    call("ivit_attention_fused_rowlut", P(ws["q"]), P(ws["k"]), P(ws["vt"]), _dy(hc[p + "attn.dy_qk"]),
                         f32[p + "attn.s_softmax"], P(self.rowtab[i]), int(hc[p + "attn.exp_meta"][2]),
                         _dy(hc[p + "attn.dy_pv"]), P(ws["ctx8"]), B, H, T, dh, ldv)
                elif p + "attn.exp_meta" in hc and self.use_exp_tables:
                    meta = hc[p + "attn.exp_meta"]
                    call("ivit_attention_fused_lut", P(ws["q"]), P(ws["k"]), P(ws["vt"]), _dy(hc[p + "attn.dy_qk"]),
                         f32[p + "attn.s_softmax"], self.ptr(p + "attn.exp_aq"), self.ptr(p + "attn.exp_t"),
                         self.ptr(p + "attn.exp_cls"), int(meta[0]), int(meta[1]), int(meta[2]),
                         _dy(hc[p + "attn.dy_pv"]), P(ws["ctx8"]), B, H, T, dh, ld)
                else:
                    call("ivit_attention_fused", P(ws["q"]), P(ws["k"]), P(ws["vt"]), _dy(hc[p + "attn.dy_qk"]),
                         f32[p + "attn.s_softmax"], _dy(hc[p + "attn.dy_pv"]), P(ws["ctx8"]), B, H, T, dh, ld)
            else:
                call("ivit_attn_qk_requant", P(ws["q"]), P(ws["k"]), _dy(hc[p + "attn.dy_qk"]), P(ws["s8"]),
                     B * H, T, dh, ld)
                call("ivit_shiftmax", P(ws["s8"]), B * H * T, T, ld, f32[p + "attn.s_softmax"], 16, P(ws["p16"]), ld)
                call("ivit_attn_pv_requant", P(ws["p16"]), P(ws["vt"]), _dy(hc[p + "attn.dy_pv"]), P(ws["ctx8"]),
                     B, H, T, dh, ld, ld)
            if self.use_plans:
                call("ivit_linear_i8_requant_residual_planned", self.plan(p + "attn.proj"), P(ws["ctx8"]),
                     _dy(hc[p + "res1.dy_main"]), _dy(hc[p + "res1.dy_res"]), P(x), P(y), M)
            else:
                call("ivit_linear_i8_requant_residual", P(ws["ctx8"]), self.ptr(p + "attn.proj.w"), self.ptr(p + "attn.proj.b"),
                     self.ptr(p + "attn.proj.dy"), _dy(hc[p + "res1.dy_main"]), _dy(hc[p + "res1.dy_res"]), P(x), P(y), M, D, D)
            x, y = y, x
            dm, dr = hc[p + "res2.dy_main"], hc[p + "res2.dy_res"]
            fused = (self.use_plans and self.use_fused_mlp and i in self._mlp_plans
                     and abs(dm[0, 0] * dm[0, 1]) < 512.0 and abs(dr[0, 0] * dr[0, 1]) < 512.0)
            # the runner's choice: norm2 in the head of the fused Mlp's launch where that launch runs on the role-split kernel
            ln_mlp = fused and self.fuse_ln_mlp and self.h.try_call(
                "ivit_layernorm_mlp_fused_planned", self._mlp_plans[i], P(x), f32[p + "ln2.s"], self.ptr(p + "norm2.bias_int"), self.ptr(p + "norm2.sc"),
                self.ptr(p + "norm2.dy"), P(ws["a8"]), _P(self.gelu_tab[i].data_ptr()), _dy(dm), _dy(dr), P(y), M)
            if not ln_mlp:
                call("ivit_layernorm_requant", P(x), M, D, D, f32[p + "ln2.s"], self.ptr(p + "norm2.bias_int"),
                     self.ptr(p + "norm2.sc"), self.ptr(p + "norm2.dy"), P(ws["a8"]))
            if ln_mlp:
                pass
            elif fused:       # hidden tensor never in HBM
                call("ivit_mlp_fused_planned", self._mlp_plans[i], P(ws["a8"]), _P(self.gelu_tab[i].data_ptr()), _dy(dm), _dy(dr),
                     P(x), P(y), M)
            elif self.use_plans:
                call("ivit_linear_i8_requant_planned", self.plan(p + "mlp.fc1"), P(ws["a8"]), 8, P(ws["h8"]), M)
                call("ivit_shiftgelu_requant_lut", P(ws["h8"]), M, Hd, _P(self.gelu_tab[i].data_ptr()), P(ws["g8"]))
                call("ivit_linear_i8_requant_residual_planned", self.plan(p + "mlp.fc2"), P(ws["g8"]), _dy(dm), _dy(dr), P(x), P(y), M)
            else:
                call("ivit_linear_i8_requant", P(ws["a8"]), self.ptr(p + "mlp.fc1.w"), self.ptr(p + "mlp.fc1.b"),
                     self.ptr(p + "mlp.fc1.dy"), 8, P(ws["h8"]), M, Hd, D)
                call("ivit_shiftgelu_requant_lut", P(ws["h8"]), M, Hd, _P(self.gelu_tab[i].data_ptr()), P(ws["g8"]))
                call("ivit_linear_i8_requant_residual", P(ws["g8"]), self.ptr(p + "mlp.fc2.w"), self.ptr(p + "mlp.fc2.b"),
                     self.ptr(p + "mlp.fc2.dy"), _dy(dm), _dy(dr), P(x), P(y), M, D, Hd)
            x, y = y, x
        # final norm on the class-token rows only (row stride T*D)
        call("ivit_layernorm_requant", P(x), B, D, T * D, f32["ln.s"], self.ptr("norm.bias_int"),
             self.ptr("norm.sc"), self.ptr("norm.dy"), P(ws["cls8"]))
        call("ivit_linear_i8", P(ws["cls8"]), self.ptr("head.w"), self.ptr("head.b"), P(ws["logits"]),
             B, cfg.num_classes, D)
        self.last_x = x
        return ws["logits"]
