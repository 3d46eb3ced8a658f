"""ViTEngine — frozen integer DeiT/ViT forward on one MI355X through the C-ABI.

Mirrors the call order of the reference `VisionTransformer.forward`
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
        self.host = {}
        # scalar dyadics are passed by value: keep host copies
        if isinstance(blob, np.ndarray):
            hb = blob
        else:
            hb = self.blob.cpu().numpy()
        for k, (o, dt, shp) in table.items():
            if dt == "<f8" and shp == (1, 2):
                self.host[k] = hb[o:o + 16].view(np.float64).reshape(1, 2).copy()
        self.h = _lib.Handle(self.device.index or 0, torch.cuda.current_stream(self.device).cuda_stream)
        self._ws = {}
        self.fused_attention = (cfg.head_dim == 64 and cfg.num_tokens <= 640)
        # per-layer ShiftGELU(+requant) tables, built on-device by the faithful kernel code
        self.gelu_tab = torch.empty(cfg.depth, 65536, dtype=torch.int8, device=self.device)
        for i in range(cfg.depth):
            p = f"blocks.{i}."
            self.h.call("ivit_shiftgelu_build_table", self.f32[p + "mlp.s_gelu"], _dy(self.host[p + "mlp.dy_gelu"]),
                        _P(self.gelu_tab[i].data_ptr()))

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

    # ------------------------------------------------------------------ execution modes
    def forward_streams(self, images, nstreams=2):
        """Split the batch into `nstreams` independent slices, each on its own HIP stream, so that
        the VALU-bound kernels of one slice (attention, LayerNorm, GELU table) can share the chip
        with the MFMA-bound GEMMs of another.  Same integers as forward()."""
        B = images.shape[0]
        if not hasattr(self, "_streams") or len(self._streams) != nstreams:
            self._streams = [torch.cuda.Stream(self.device) for _ in range(nstreams)]
        cur = torch.cuda.current_stream(self.device)
        bounds = [(B * i) // nstreams for i in range(nstreams + 1)]
        outs = []
        for i, st in enumerate(self._streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(self.forward(images[bounds[i]:bounds[i + 1]], ws_key=("s", i)))
        for st in self._streams:
            cur.wait_stream(st)
        return torch.cat(outs, 0)

    def capture(self, images, nstreams=1):
        """hipGraph capture of one forward (all kernels are capturable: no allocation / sync).
        Returns a callable replaying the graph on the same `images` buffer -> logits tensor."""
        run = (lambda: self.forward(images)) if nstreams <= 1 else (lambda: self.forward_streams(images, nstreams))
        run()                                   # warm up: allocates workspaces outside the capture
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = run()

        def replay():
            g.replay()
            return out
        return replay

    def forward(self, images, ws_key=None):
        """images: int8 device tensor [B, C, H, W] (already quantised, scale s_in).
        Returns int32 logits [B, num_classes] (head accumulators)."""
        cfg, call, f32, hc = self.cfg, self.h.call, self.f32, self.host
        assert images.dtype == torch.int8 and images.is_contiguous() and images.device == self.device
        self.h.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        B = images.shape[0]
        T, D, H, dh, Hd = cfg.num_tokens, cfg.embed_dim, cfg.num_heads, cfg.head_dim, cfg.hidden_dim
        M = B * T
        ws = self.workspace(B, ws_key)
        ld = ws["ld"]
        P = lambda t: _P(t.data_ptr())
        Kp = cfg.in_chans * cfg.patch_size ** 2
        call("ivit_im2col_patch", P(images), B, cfg.in_chans, cfg.img_size, cfg.img_size, cfg.patch_size,
             P(ws["patches"]))
        call("ivit_linear_i8_requant", P(ws["patches"]), self.ptr("patch_embed.proj.w"),
             self.ptr("patch_embed.proj.b"), self.ptr("patch_embed.proj.dy"), 16, P(ws["patch16"]),
             B * cfg.num_patches, D, Kp)
        x, y = ws["xa"], ws["xb"]
        call("ivit_embed_finish", P(ws["patch16"]), self.ptr("z_cls"), self.ptr("pos"), _dy(hc["embed.dy_x"]),
             _dy(hc["embed.dy_pos"]), P(x), B, T, D)
        for i in range(cfg.depth):
            p = f"blocks.{i}."
            call("ivit_layernorm_requant", P(x), M, D, D, f32[p + "ln1.s"], self.ptr(p + "norm1.bias_int"),
                 self.ptr(p + "norm1.sc"), self.ptr(p + "norm1.dy"), P(ws["a8"]))
            call("ivit_linear_i8_qkv", P(ws["a8"]), self.ptr(p + "attn.qkv.w"), self.ptr(p + "attn.qkv.b"),
                 self.ptr(p + "attn.qkv.dy"), P(ws["q"]), P(ws["k"]), P(ws["vt"]), B, T, H, dh, ld)
            if self.fused_attention:
                call("ivit_attention_fused", P(ws["q"]), P(ws["k"]), P(ws["vt"]), _dy(hc[p + "attn.dy_qk"]),
                     f32[p + "attn.s_softmax"], _dy(hc[p + "attn.dy_pv"]), P(ws["ctx8"]), B, H, T, dh, ld)
            else:
                call("ivit_attn_qk_requant", P(ws["q"]), P(ws["k"]), _dy(hc[p + "attn.dy_qk"]), P(ws["s8"]),
                     B * H, T, dh, ld)
                call("ivit_shiftmax", P(ws["s8"]), B * H * T, T, ld, f32[p + "attn.s_softmax"], 16, P(ws["p16"]), ld)
                call("ivit_attn_pv_requant", P(ws["p16"]), P(ws["vt"]), _dy(hc[p + "attn.dy_pv"]), P(ws["ctx8"]),
                     B, H, T, dh, ld, ld)
            call("ivit_linear_i8_requant_residual", P(ws["ctx8"]), self.ptr(p + "attn.proj.w"),
                 self.ptr(p + "attn.proj.b"), self.ptr(p + "attn.proj.dy"), _dy(hc[p + "res1.dy_main"]),
                 _dy(hc[p + "res1.dy_res"]), P(x), P(y), M, D, D)
            x, y = y, x
            call("ivit_layernorm_requant", P(x), M, D, D, f32[p + "ln2.s"], self.ptr(p + "norm2.bias_int"),
                 self.ptr(p + "norm2.sc"), self.ptr(p + "norm2.dy"), P(ws["a8"]))
            call("ivit_linear_i8_requant", P(ws["a8"]), self.ptr(p + "mlp.fc1.w"), self.ptr(p + "mlp.fc1.b"),
                 self.ptr(p + "mlp.fc1.dy"), 8, P(ws["h8"]), M, Hd, D)
            call("ivit_shiftgelu_requant_lut", P(ws["h8"]), M, Hd, _P(self.gelu_tab[i].data_ptr()), P(ws["g8"]))
            call("ivit_linear_i8_requant_residual", P(ws["g8"]), self.ptr(p + "mlp.fc2.w"),
                 self.ptr(p + "mlp.fc2.b"), self.ptr(p + "mlp.fc2.dy"), _dy(hc[p + "res2.dy_main"]),
                 _dy(hc[p + "res2.dy_res"]), P(x), P(y), M, D, Hd)
            x, y = y, x
        # final norm on the class-token rows only (row stride T*D)
        call("ivit_layernorm_requant", P(x), B, D, T * D, f32["ln.s"], self.ptr("norm.bias_int"),
             self.ptr("norm.sc"), self.ptr("norm.dy"), P(ws["cls8"]))
        call("ivit_linear_i8", P(ws["cls8"]), self.ptr("head.w"), self.ptr("head.b"), P(ws["logits"]),
             B, cfg.num_classes, D)
        self.last_x = x
        return ws["logits"]
