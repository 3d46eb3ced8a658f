"""SwinEngine — frozen integer Swin forward on one MI355X through the C-ABI.

Call order of the reference `SwinTransformer.forward_features/forward`
(models/swin_quant.py:539-564; SwinTransformerBlock :251-301; WindowAttention :121-169;
PatchMerging :328-349; PatchEmbed layers_quant.py:184-196) on the fused kernels:

  * activations stay in NATURAL token order [B, R*R, C] for the whole network; torch.roll,
    window_partition and window_reverse exist only as index arithmetic inside
    `ivit_window_attention_fused`;
  * QuantLinear -> QuantAct, IntLayerNorm -> QuantAct, IntGELU -> QuantAct and
    QuantAct -> QuantAct(identity) pairs are single kernels, as in the ViT engine;
  * stage-0 LayerNorms use torch's token-contiguous summation order.

Constants are derived once on the host with the reference's fp32/fp64 operation order
(`freeze_swin`).  torch is used for device memory and streams only.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .engine import pack_constants
from .freeze import dyadic, layernorm_constants, quantize, quantize_bias, quantize_weight, shiftmax_tables

_P = ctypes.c_void_p


def _rel_index(ws):
    """relative_position_index (swin_quant.py:80-94)"""
    coords = np.stack(np.meshgrid(np.arange(ws), np.arange(ws), indexing="ij")).reshape(2, -1)
    rel = (coords[:, :, None] - coords[:, None, :]).transpose(1, 2, 0).copy()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def _rne_times(z, dy):
    """rne((double(z) * m) * 2^-e) for an integer array and one dyadic pair"""
    return np.rint(z.astype(np.float64) * dy[0, 0] * dy[0, 1])


def freeze_swin(cfg, weights, scales, exp_tables=False):
    """name -> numpy array / python scalar for every constant of the frozen Swin.
    exp_tables: also build the Shiftmax tables of every layer, which routes the windowed attention to
    ivit_window_attention_fused_lut.  Off by default: measured on MI355X the table form is not faster than the
    arithmetic one in this kernel (profiles/README.md, round 4) — the switch keeps the path exercised by the tests."""
    s = {k: np.float32(v) for k, v in scales.items()}
    c = {}

    def linear(prefix, s_in, s_out_site):
        wq, s_w = quantize_weight(weights[prefix + ".weight"])
        c[prefix + ".w"] = np.ascontiguousarray(wq.reshape(wq.shape[0], -1))
        if prefix + ".bias" in weights:
            bq, s_b = quantize_bias(weights[prefix + ".bias"], s_w, s_in)
            c[prefix + ".b"] = bq
        else:
            s_b = (s_w * np.float32(s_in)).astype(np.float32)
        if s_out_site is not None:
            c[prefix + ".dy"] = dyadic(s_b, s[s_out_site])
        return s_b

    def norm(prefix, s_out_site):
        bi, sc = layernorm_constants(weights[prefix + ".weight"], weights[prefix + ".bias"])
        c[prefix + ".bias_int"], c[prefix + ".sc"] = bi, sc
        c[prefix + ".dy"] = dyadic(sc, s[s_out_site])

    linear("patch_embed.proj", s["qact_input"], "patch_embed.qact_before_norm")
    c["patch_embed.s_bn"] = s["patch_embed.qact_before_norm"]
    norm("patch_embed.norm", "patch_embed.qact")
    c["dy_qact1"] = dyadic(s["patch_embed.qact"], s["qact1"])
    s_x = s["qact1"]
    res = cfg.grid
    for li, (depth, heads) in enumerate(zip(cfg.depths, cfg.num_heads)):
        C = cfg.embed_dim * 2 ** li
        dh = C // heads
        ws = min(cfg.window_size, res)
        for bj in range(depth):
            p = f"layers.{li}.blocks.{bj}."
            c[p + "s_in"] = s_x
            norm(p + "norm1", p + "qact1")
            linear(p + "attn.qkv", s[p + "qact1"], p + "attn.qact1")
            s1 = s[p + "attn.qact1"]
            s_qk = np.float32(np.float32(s1 * s1) * np.float32(dh ** -0.5))       # swin_quant.py:133-135
            c[p + "attn.dy_qk"] = dyadic(s_qk, s[p + "attn.qact_attn1"])
            c[p + "attn.dy_a"] = dyadic(s[p + "attn.qact_attn1"], s[p + "attn.qact2"])
            # relative position bias: table -> 8 bit (QuantAct input branch), gathered, then its half of
            # the identity requant of qact2 (swin_quant.py:142-149) — a constant of the layer
            tab = quantize(weights[p + "attn.relative_position_bias_table"], s[p + "attn.qact_table"], 8, False)
            N = ws * ws
            bias = tab[_rel_index(ws).reshape(-1)].reshape(N, N, heads).transpose(2, 0, 1)
            c[p + "attn.relb"] = np.ascontiguousarray(
                _rne_times(bias, dyadic(s[p + "attn.qact_table"], s[p + "attn.qact2"]))).astype(np.int16)
            c[p + "attn.s_softmax"] = s[p + "attn.qact2"]
            tabs = shiftmax_tables(s[p + "attn.qact2"]) if exp_tables else None   # exp_int by table where no shift mask applies
            if tabs is not None:                               # else: the kernel's arithmetic path for this layer
                c[p + "attn.exp_aq"], c[p + "attn.exp_t"], c[p + "attn.exp_cls"] = tabs["aq"], tabs["t"], tabs["cls"]
                # three small integers, carried with the fp32 host scalars (exact: t_count <= 16384)
                c[p + "attn.exp_nc"], c[p + "attn.exp_tcount"], c[p + "attn.exp_dmin"] = tabs["NC"], tabs["t"].size, tabs["dmin"]
            c[p + "attn.dy_pv"] = dyadic(np.float32(np.float32(2.0 ** -7) * s1), s[p + "attn.qact3"])
            linear(p + "attn.proj", s[p + "attn.qact3"], p + "attn.qact4")
            c[p + "res1.dy_main"] = dyadic(s[p + "attn.qact4"], s[p + "qact2"])
            c[p + "res1.dy_res"] = dyadic(s_x, s[p + "qact2"])
            c[p + "s_mid"] = s[p + "qact2"]
            norm(p + "norm2", p + "qact3")
            linear(p + "mlp.fc1", s[p + "qact3"], p + "mlp.qact_gelu")
            c[p + "mlp.s_gelu"] = s[p + "mlp.qact_gelu"]
            c[p + "mlp.dy_gelu"] = dyadic(np.float32(s[p + "mlp.qact_gelu"] * np.float32(2.0 ** -7)), s[p + "mlp.qact1"])
            linear(p + "mlp.fc2", s[p + "mlp.qact1"], p + "mlp.qact2")
            c[p + "res2.dy_main"] = dyadic(s[p + "mlp.qact2"], s[p + "qact4"])
            c[p + "res2.dy_res"] = dyadic(s[p + "qact2"], s[p + "qact4"])
            s_x = s[p + "qact4"]
        if li < cfg.num_layers - 1:
            p = f"layers.{li}.downsample."
            c[p + "s_in"] = s_x
            norm(p + "norm", p + "qact1")
            linear(p + "reduction", s[p + "qact1"], p + "qact2")
            s_x = s[p + "qact2"]
            res //= 2
    c["norm.s_in"] = s_x
    norm("norm", "qact2")
    c["dy_pool"] = dyadic(s["qact2"], s["qact3"])
    c["head.scale"] = linear("head", s["qact3"], None)
    return c


DEVICE_DYADICS = ("dy_qact1",)          # [1,2] tables the kernels read through a pointer


def pack_swin_constants(consts):
    """freeze_swin output -> (byte blob, table, host scalars): the unit one RCCL broadcast carries.
    Arrays go into the blob; by-value dyadics and fp32 scalars travel in `host` (a small picklable dict)."""
    arrays, host = {}, {}
    for k, v in consts.items():
        if isinstance(v, np.ndarray) and v.dtype == np.float64 and v.shape == (1, 2) and k not in DEVICE_DYADICS:
            host[k] = ("dy", float(v[0, 0]), float(v[0, 1]))
        elif isinstance(v, np.ndarray) and v.ndim >= 1:
            arrays[k] = v
        else:
            host[k] = ("f", float(np.float32(v)))
    blob, table = pack_constants(arrays)
    return blob, table, host


def swin_host_scalars(host):
    """(fp32 scalars, by-value dyadics) from the host part of pack_swin_constants"""
    return ({k: v[1] for k, v in host.items() if v[0] == "f"},
            {k: _lib.Dyadic(v[1], v[2]) for k, v in host.items() if v[0] == "dy"})


def swin_native_params(cfg, table, f, dy, base):
    """(ivit_swin_config, ivit_swin_params, keep-alive) for ivit_swin_create — or its CPU twin: `base` is the address of the
    packed constants blob (device memory for the library, host memory for oracle/ivit_twin.c)"""
    L = _lib
    a = lambda name: (base + table[name][0]) if name in table else None
    ln = lambda p: L.LnParams(a(p + ".bias_int"), a(p + ".sc"), a(p + ".dy"))
    lin = lambda p: L.LinParams(a(p + ".w"), a(p + ".b"), a(p + ".dy"))
    nb = sum(cfg.depths)
    blocks = (L.SwinBlock * nb)()
    merges = (L.SwinMerge * max(1, cfg.num_layers - 1))()
    i = 0
    for li, depth in enumerate(cfg.depths):
        for bj in range(depth):
            p, b = f"layers.{li}.blocks.{bj}.", blocks[i]
            b.s_in, b.n1, b.qkv = f[p + "s_in"], ln(p + "norm1"), lin(p + "attn.qkv")
            b.dy_qk, b.dy_a, b.relb = dy[p + "attn.dy_qk"], dy[p + "attn.dy_a"], a(p + "attn.relb")
            b.s_softmax, b.dy_pv, b.proj = f[p + "attn.s_softmax"], dy[p + "attn.dy_pv"], lin(p + "attn.proj")
            if p + "attn.exp_aq" in table:
                b.exp_aq, b.exp_t, b.exp_cls = a(p + "attn.exp_aq"), a(p + "attn.exp_t"), a(p + "attn.exp_cls")
                b.exp_nc, b.exp_tcount, b.exp_dmin = (int(f[p + "attn.exp_nc"]), int(f[p + "attn.exp_tcount"]),
                                                      int(f[p + "attn.exp_dmin"]))
            b.res1_main, b.res1_res = dy[p + "res1.dy_main"], dy[p + "res1.dy_res"]
            b.s_mid, b.n2, b.fc1 = f[p + "s_mid"], ln(p + "norm2"), lin(p + "mlp.fc1")
            b.s_gelu, b.dy_gelu, b.fc2 = f[p + "mlp.s_gelu"], dy[p + "mlp.dy_gelu"], lin(p + "mlp.fc2")
            b.res2_main, b.res2_res = dy[p + "res2.dy_main"], dy[p + "res2.dy_res"]
            i += 1
        if li < cfg.num_layers - 1:
            p, g = f"layers.{li}.downsample.", merges[li]
            g.s_in, g.n, g.red = f[p + "s_in"], ln(p + "norm"), lin(p + "reduction")
    prm = L.SwinParams()
    prm.pe, prm.s_bn, prm.pn, prm.dy_qact1 = lin("patch_embed.proj"), f["patch_embed.s_bn"], ln("patch_embed.norm"), a("dy_qact1")
    prm.blocks_host = ctypes.cast(blocks, ctypes.POINTER(L.SwinBlock))
    prm.merges_host = ctypes.cast(merges, ctypes.POINTER(L.SwinMerge))
    prm.s_norm_in, prm.n, prm.dy_pool = f["norm.s_in"], ln("norm"), dy["dy_pool"]
    prm.head_w, prm.head_b = a("head.w"), a("head.b")
    c = L.SwinConfigC(cfg.img_size, cfg.patch_size, cfg.in_chans, cfg.embed_dim, cfg.num_layers, cfg.window_size,
                      int(cfg.mlp_ratio), cfg.num_classes, (ctypes.c_int * 4)(*(list(cfg.depths) + [0] * 4)[:4]),
                      (ctypes.c_int * 4)(*(list(cfg.num_heads) + [0] * 4)[:4]))
    return c, prm, (blocks, merges)


class SwinEngine:
    def __init__(self, cfg, weights, scales, device="cuda:0", packed=None, exp_tables=False):
        """weights/scales: freeze here (rank 0) — or `packed` = (blob, table, host) received from a broadcast.
        exp_tables: see freeze_swin."""
        if not torch.cuda.is_available():
            raise _lib.IvitError("SwinEngine needs a HIP device; the product path has no CPU fallback")
        self.cfg, self.device = cfg, torch.device(device)
        torch.cuda.set_device(self.device)
        if cfg.window_size != 7 or any((cfg.embed_dim * 2 ** i) // h != 32 for i, h in enumerate(cfg.num_heads)):
            raise _lib.IvitError("the fused windowed attention is built for window 7 / head dim 32")
        blob, table, host = packed if packed is not None else pack_swin_constants(freeze_swin(cfg, weights, scales, exp_tables))
        self.table, self.host_consts = table, host
        self.blob = torch.from_numpy(blob).to(self.device) if isinstance(blob, np.ndarray) else blob.to(self.device)
        o, dt, shp = table["head.scale"]
        self.head_scale = self.blob[o:o + 4 * int(np.prod(shp))].cpu().numpy().view(np.float32).copy()
        self.f, self.dy = swin_host_scalars(host)
        self.t = _BlobView(self.blob, table)
        self.h = _lib.Handle(self.device.index if self.device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(self.device).cuda_stream)
        self._build_native()
        # per-layer ShiftGELU(+requant) tables for forward_ops (the native runner owns its own copies)
        cfg = self.cfg
        self.gelu = {}
        for li, depth in enumerate(cfg.depths):
            for bj in range(depth):
                p = f"layers.{li}.blocks.{bj}."
                tab = torch.empty(65536, dtype=torch.int8, device=self.device)
                self.h.call("ivit_shiftgelu_build_table", self.f[p + "mlp.s_gelu"], self.dy[p + "mlp.dy_gelu"], _P(tab.data_ptr()))
                self.gelu[p] = tab
        self._ws = {}
        self.use_exp_tables = True      # forward_ops only: False issues the arithmetic Shiftmax in every window (cross-check)

    MAX_SLICES = 8

    def ptr(self, name):
        return _P(self.t.addr(name))

    def _dyp(self, name):
        return self.ptr(name)

    def _build_native(self):
        """ivit_swin_create: device pointers into the blob + host scalars -> one C call per batch"""
        c, prm, self._native_keep = swin_native_params(self.cfg, self.table, self.f, self.dy, self.blob.data_ptr())
        self.model = _P()
        self.h._check(self.h.lib.ivit_swin_create(self.h.h, ctypes.byref(c), ctypes.byref(prm), self.MAX_SLICES,
                                                  ctypes.byref(self.model)), "ivit_swin_create")
        self._native_ws = {}

    def __del__(self):
        try:
            if getattr(self, "model", None):
                self.h.lib.ivit_swin_destroy(self.model)
                self.model = None
        except Exception:
            pass

    def _native_buffers(self, B, nslices):
        key = (B, nslices)
        if key not in self._native_ws:
            n = ctypes.c_size_t()
            self.h._check(self.h.lib.ivit_swin_workspace_bytes(self.model, B, nslices, ctypes.byref(n)), "ivit_swin_workspace_bytes")
            self._native_ws[key] = (torch.empty(n.value, dtype=torch.uint8, device=self.device),
                                    torch.empty(B, self.cfg.num_classes, dtype=torch.int32, device=self.device))
        return self._native_ws[key]

    def workspace(self, B, key=None):
        if (B, key) in self._ws:
            return self._ws[(B, key)]
        cfg, dev = self.cfg, self.device
        L0, E = cfg.grid * cfg.grid, cfg.embed_dim
        e = lambda n, dt: torch.empty(n, dtype=dt, device=dev)
        M0 = B * L0
        ws = dict(
            patches=e(M0 * cfg.in_chans * cfg.patch_size ** 2, torch.int8),
            a8=e(M0 * E, torch.int8),            # LN output / generic int8 [M, C]
            x16a=e(M0 * E, torch.int16), x16b=e(M0 * E, torch.int16), x16c=e(M0 * E, torch.int16),
            zf=e(M0 * E, torch.float32),
            qkv=e(M0 * 3 * E, torch.int8),
            ctx=e(M0 * E, torch.int8),
            h8=e(M0 * 4 * E, torch.int8), g8=e(M0 * 4 * E, torch.int8),
            pool=e(B * E * 2 ** (cfg.num_layers - 1), torch.int8),
            logits=torch.empty(B, cfg.num_classes, dtype=torch.int32, device=dev),
        )
        self._ws[(B, key)] = ws
        return ws

    def forward(self, images, nslices=1):
        """images int8 [B, C, H, W] (scale qact_input) -> int32 logits [B, num_classes].
        nslices > 1: independent batch slices on separate HIP streams (VALU-bound attention / LayerNorm of
        one slice share the chip with the GEMMs of another); same integers."""
        assert images.dtype == torch.int8 and images.is_contiguous() and images.device == self.device
        self.h.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        B = images.shape[0]
        nslices = max(1, min(int(nslices), B, self.MAX_SLICES))
        ws, logits = self._native_buffers(B, nslices)
        self.h._check(self.h.lib.ivit_swin_forward(self.model, _P(images.data_ptr()), B, nslices, _P(ws.data_ptr()),
                                                   ws.numel(), _P(logits.data_ptr())), "ivit_swin_forward")
        return logits

    def forward_ops(self, images, nslices=1):
        """the same forward issued one C-ABI call per operator from Python (per-operator timing)"""
        if nslices > 1 and images.shape[0] >= nslices:
            return self._forward_sliced(images, nslices)
        return self._forward_one(images, None)

    def _forward_sliced(self, images, nslices):
        B = images.shape[0]
        if not hasattr(self, "_streams") or len(self._streams) != nslices:
            self._streams = [torch.cuda.Stream(self.device) for _ in range(nslices)]
        cur = torch.cuda.current_stream(self.device)
        bounds = [(B * i) // nslices for i in range(nslices + 1)]
        outs = []
        for i, st in enumerate(self._streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(self._forward_one(images[bounds[i]:bounds[i + 1]], ("slice", i)))
        for st in self._streams:
            cur.wait_stream(st)
        return torch.cat(outs, 0)

    def _forward_one(self, images, ws_key):
        cfg, call, f, dy = self.cfg, self.h.call, self.f, self.dy
        assert images.dtype == torch.int8 and images.is_contiguous() and images.device == self.device
        self.h.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        B = images.shape[0]
        ws = self.workspace(B, ws_key)
        P = lambda t: _P(t.data_ptr())
        E, res = cfg.embed_dim, cfg.grid
        L = res * res
        M = B * L
        Kp = cfg.in_chans * cfg.patch_size ** 2
        # ---- PatchEmbed: conv(4x4/4) -> qact_before_norm(8) -> norm -> qact(16) -> qact1(16)
        call("ivit_im2col_patch", P(images), B, cfg.in_chans, cfg.img_size, cfg.img_size, cfg.patch_size, P(ws["patches"]))
        call("ivit_linear_i8_requant", P(ws["patches"]), self.ptr("patch_embed.proj.w"), self.ptr("patch_embed.proj.b"),
             self.ptr("patch_embed.proj.dy"), 8, P(ws["a8"]), M, E, Kp)
        call("ivit_widen_i8_i16", P(ws["a8"]), P(ws["x16b"]), M * E)
        call("ivit_layernorm_tokenorder", P(ws["x16b"]), M, E, f["patch_embed.s_bn"],
             self.ptr("patch_embed.norm.bias_int"), self.ptr("patch_embed.norm.sc"), L, P(ws["zf"]))
        call("ivit_requant_f32", P(ws["zf"]), self.ptr("patch_embed.norm.dy"), E, None, None, 16, P(ws["x16b"]), M, E)
        call("ivit_requant_i16", P(ws["x16b"]), self._dyp("dy_qact1"), 1, None, None, 16, P(ws["x16a"]), M, E)
        x, y, t16 = ws["x16a"], ws["x16b"], ws["x16c"]
        for li, (depth, heads) in enumerate(zip(cfg.depths, cfg.num_heads)):
            C = E * 2 ** li
            for bj in range(depth):
                p = f"layers.{li}.blocks.{bj}."
                shift = 0 if (bj % 2 == 0 or res <= cfg.window_size) else cfg.window_size // 2
                self._ln(x, M, C, f[p + "s_in"], p + "norm1", L, li == 0, ws["a8"])
                call("ivit_linear_i8_requant", P(ws["a8"]), self.ptr(p + "attn.qkv.w"), self.ptr(p + "attn.qkv.b"),
                     self.ptr(p + "attn.qkv.dy"), 8, P(ws["qkv"]), M, 3 * C, C)
                if p + "attn.exp_aq" in self.table and self.use_exp_tables:
                    call("ivit_window_attention_fused_lut", P(ws["qkv"]), dy[p + "attn.dy_qk"], dy[p + "attn.dy_a"],
                         self.ptr(p + "attn.relb"), f[p + "attn.s_softmax"], self.ptr(p + "attn.exp_aq"),
                         self.ptr(p + "attn.exp_t"), self.ptr(p + "attn.exp_cls"), int(f[p + "attn.exp_nc"]),
                         int(f[p + "attn.exp_tcount"]), int(f[p + "attn.exp_dmin"]), dy[p + "attn.dy_pv"], P(ws["ctx"]),
                         B, res, cfg.window_size, shift, heads, C // heads)
                else:
                    call("ivit_window_attention_fused", P(ws["qkv"]), dy[p + "attn.dy_qk"], dy[p + "attn.dy_a"],
                         self.ptr(p + "attn.relb"), f[p + "attn.s_softmax"], dy[p + "attn.dy_pv"], P(ws["ctx"]),
                         B, res, cfg.window_size, shift, heads, C // heads)
                call("ivit_linear_i8_requant_residual", P(ws["ctx"]), self.ptr(p + "attn.proj.w"), self.ptr(p + "attn.proj.b"),
                     self.ptr(p + "attn.proj.dy"), dy[p + "res1.dy_main"], dy[p + "res1.dy_res"], P(x), P(y), M, C, C)
                x, y = y, x
                self._ln(x, M, C, f[p + "s_mid"], p + "norm2", L, li == 0, ws["a8"])
                call("ivit_linear_i8_requant", P(ws["a8"]), self.ptr(p + "mlp.fc1.w"), self.ptr(p + "mlp.fc1.b"),
                     self.ptr(p + "mlp.fc1.dy"), 8, P(ws["h8"]), M, 4 * C, C)
                call("ivit_shiftgelu_requant_lut", P(ws["h8"]), M, 4 * C, P(self.gelu[p]), P(ws["g8"]))
                call("ivit_linear_i8_requant_residual", P(ws["g8"]), self.ptr(p + "mlp.fc2.w"), self.ptr(p + "mlp.fc2.b"),
                     self.ptr(p + "mlp.fc2.dy"), dy[p + "res2.dy_main"], dy[p + "res2.dy_res"], P(x), P(y), M, C, 4 * C)
                x, y = y, x
            if li < cfg.num_layers - 1:     # PatchMerging: gather -> LN(4C) -> qact1(8) -> reduction -> qact2(8)
                p = f"layers.{li}.downsample."
                call("ivit_patch_merge_gather", P(x), 16, B, res, C, P(t16))
                res //= 2
                L = res * res
                M = B * L
                self._ln(t16, M, 4 * C, f[p + "s_in"], p + "norm", L, False, ws["a8"])
                call("ivit_linear_i8_requant", P(ws["a8"]), self.ptr(p + "reduction.w"), None,
                     self.ptr(p + "reduction.dy"), 8, P(ws["ctx"]), M, 2 * C, 4 * C)
                call("ivit_widen_i8_i16", P(ws["ctx"]), P(x), M * 2 * C)
        C = E * 2 ** (cfg.num_layers - 1)
        self._ln(x, M, C, f["norm.s_in"], "norm", L, False, ws["a8"])
        call("ivit_avgpool_requant", P(ws["a8"]), B, L, C, dy["dy_pool"], P(ws["pool"]))
        call("ivit_linear_i8", P(ws["pool"]), self.ptr("head.w"), self.ptr("head.b"), P(ws["logits"]), B, cfg.num_classes, C)
        return ws["logits"]

    def capture(self, images, nslices=1):
        """hipGraph of one forward on fixed buffers (ivit_swin_graph_create); returns a replay callable."""
        B = images.shape[0]
        nslices = max(1, min(int(nslices), B, self.MAX_SLICES))
        ws, logits = self._native_buffers(B, nslices)
        if not hasattr(self, "_gstream"):
            self._gstream = torch.cuda.Stream(self.device)
        torch.cuda.synchronize(self.device)
        self.h.set_stream(self._gstream.cuda_stream)
        g = _P()
        self.h._check(self.h.lib.ivit_swin_graph_create(self.model, _P(images.data_ptr()), B, nslices, _P(ws.data_ptr()),
                                                        ws.numel(), _P(logits.data_ptr()), ctypes.byref(g)), "ivit_swin_graph_create")
        self._graphs = getattr(self, "_graphs", []) + [(g, ws, logits, images)]      # the graph's buffers live as long as it does
        lib, gs, dev = self.h.lib, self._gstream, self.device

        def replay(_keep=(ws, logits, images)):
            cur = torch.cuda.current_stream(dev)
            gs.wait_stream(cur)
            self.h.set_stream(gs.cuda_stream)
            self.h._check(lib.ivit_graph_launch(g), "ivit_graph_launch")
            cur.wait_stream(gs)
            return logits
        return replay

    def _ln(self, x16, M, C, s_in, name, L, token_order, out8):
        P = lambda t: _P(t.data_ptr())
        if token_order:
            self.h.call("ivit_layernorm_tokenorder_requant", P(x16), M, C, s_in, self.ptr(name + ".bias_int"),
                        self.ptr(name + ".sc"), self.ptr(name + ".dy"), L, P(out8))
        else:
            self.h.call("ivit_layernorm_requant", P(x16), M, C, C, s_in, self.ptr(name + ".bias_int"),
                        self.ptr(name + ".sc"), self.ptr(name + ".dy"), P(out8))


class _BlobView:
    """name -> device address / tensor view inside the packed constants blob"""

    def __init__(self, blob, table):
        self.blob, self.table = blob, table

    def __contains__(self, name):
        return name in self.table

    def addr(self, name):
        return self.blob.data_ptr() + self.table[name][0]
