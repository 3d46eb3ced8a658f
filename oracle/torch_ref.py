"""torch_ref.py — PyTorch-CPU counterpart of the reference's frozen fake-quant ViT forward ("the build's
own PyTorch-CPU counterpart of the fake-quant path, same op sequence, written from the spec in Appendix A").

TEST INFRASTRUCTURE, like the rest of oracle/: only tests/ and bench.py's `cpu_baseline` leg import it.  It exists because
the reference (a Python package) never travels to the GPU box, while the north star asks for "the reference CPU PyTorch
path timed on the same box's host cores": this file is that path restated — fp32 `(X = Q*s, s)` tensors flowing through
torch CPU operators (addmm / bmm / elementwise fp32, fp64 only inside the dyadic requant) in the order of
models/vit_quant.py:254-282 (VisionTransformer.forward), :130-143 (Block), :59-88 (Attention), layers_quant.py:144-153 (Mlp)
and the operators of quantization_utils/quant_modules.py (:67-97 QuantLinear, :165-206 QuantAct, :223-228 QuantMatMul,
:353-386 IntLayerNorm, :410-445 IntGELU, :469-497 IntSoftmax), each following SURVEY.md Appendix A.3-A.6.  Because it runs
on torch CPU, torch's own summation order and the `scalar / tensor` lowering apply to it exactly as they do to the
reference: tests/test_oracle_golden.py::test_torch_ref_matches_golden_logits pins it to the reference's int32 logits.

Constants (integer weights / biases, their scales, the I-LayerNorm integers, the dyadic (m, 2^-e) pairs) come from the
numpy helpers of oracle.py, which restate quant_modules.py:68-89 and quant_utils.py:150-175 with the reference's rounding.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import oracle as orc


def _t(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a))).to(dtype)


class _Dy:
    """dyadic multiplier(s) of one requant site: out = rne(rne(x / s_pre) * m * 2^-e) (Appendix A.3)"""

    def __init__(self, s_pre, s_out):
        d = orc.dyadic(np.asarray(s_pre, np.float32), np.float32(s_out))          # per input-scale element: (m, 2^-e)
        self.m = _t(np.array([e.m for e in d], np.float64), torch.float64)
        self.r = _t(np.array([e.r for e in d], np.float64), torch.float64)
        self.s_pre = _t(np.asarray(s_pre, np.float32).reshape(-1))


def _fixedpoint(x, dy):
    z = torch.round(x / dy.s_pre).to(torch.float64)
    return torch.round((z * dy.m) * dy.r)


def quant_act(x, dy, s_out, bits, identity=None, dy_id=None):
    """QuantAct on a fake-quant tensor: dyadic requant (+ identity requant, added before the clamp), back to fp32 * s_out"""
    out = _fixedpoint(x, dy)
    if identity is not None:
        out = out + _fixedpoint(identity, dy_id)
    n = 2 ** (bits - 1) - 1
    return torch.clamp(out.to(torch.float32), -n - 1, n) * s_out


def quant_linear(x, s_x, w_int, b_int, s_b):
    return F.linear(x / s_x, w_int, b_int) * s_b


def _iexp(x, x0, n):
    """integer shift-exponential of Shiftmax / ShiftGELU (Appendix A.4, lines t .. e), x <= 0 up to the GELU's exp(-max)"""
    t = x + torch.floor(x / 2) - torch.floor(x / 2 ** 4)
    t = torch.max(t, n * x0)
    q = torch.floor(t / x0)
    r = t - x0 * q
    e = r / 2 - x0
    return torch.clamp(torch.floor(e * 2 ** (n - q)), min=0)


def int_softmax(x, s, bits):
    xi = x / s
    xi = xi - xi.max(dim=-1, keepdim=True)[0]
    e = _iexp(xi, torch.floor(-1.0 / s), 15)
    ssum = e.sum(dim=-1, keepdim=True).clamp_max(2 ** 31 - 1)
    factor = torch.floor((2 ** 31 - 1) / ssum)
    out = torch.floor(e * factor / 2 ** (31 - bits + 1))
    s_out = torch.tensor(1.0 / 2 ** (bits - 1))
    return out * s_out, s_out


def int_gelu(x, s):
    p = x / s
    ssig = s * 1.702
    x0 = torch.floor(-1.0 / ssig)
    pmax = p.max(dim=-1, keepdim=True)[0]
    e = _iexp(p - pmax, x0, 23)
    emax = _iexp(-pmax, x0, 23)
    ssum = (e + emax).clamp_max(2 ** 31 - 1)
    factor = torch.floor((2 ** 31 - 1) / ssum)
    sig = torch.floor(e * factor / 2 ** (31 - 8 + 1))
    s_out = s * torch.tensor(1.0 / 2 ** 7)
    return (p * sig) * s_out, s_out


def int_layernorm(x, s, bias_int, sc):
    """I-LayerNorm (Appendix A.6) on [B, N, C]; bias_int / sc are the frozen per-channel integers and output scales"""
    xi = x / s
    mean = torch.round(xi.mean(dim=2, keepdim=True))
    y = xi - mean
    var = torch.sum(y ** 2, dim=2, keepdim=True)
    k = torch.full_like(var, 2.0 ** 16)
    for _ in range(10):
        k = torch.floor((k + torch.floor(var / k)) / 2)
    factor = torch.floor((2 ** 31 - 1) / k)
    yi = torch.floor(y * factor / 2)
    return (yi + bias_int) * sc, sc


class TorchRefViT:
    """frozen fake-quant DeiT / ViT on torch CPU.  `weights`: float32 arrays keyed like the reference state dict;
    `scales`: {QuantAct site name -> fp32 act_scaling_factor} (the reference's calibration result)."""

    def __init__(self, cfg, weights, scales):
        self.cfg = cfg
        s = {k: np.float32(v) for k, v in scales.items()}
        self.s = {k: torch.tensor(float(v)) for k, v in s.items()}
        D = cfg.embed_dim
        self.hs = float(cfg.head_dim ** -0.5)

        def lin(wname, s_in):
            w_int, s_w = orc.weight_quant(weights[wname + ".weight"])
            b_int, s_b = orc.bias_quant(weights[wname + ".bias"], s_w, s_in)
            return _t(w_int.reshape(w_int.shape[0], -1)), _t(b_int), _t(s_b), s_b

        c = {}
        c["pe"] = lin("patch_embed.proj", s["qact_input"])
        c["pe_dy"] = _Dy(c["pe"][3], s["patch_embed.qact"])
        c["cls"] = _t(weights["cls_token"].reshape(1, 1, D))
        pos_int = orc.quantize_sym(weights["pos_embed"][0], s["qact_pos"], 16, False)
        c["pos"] = _t(pos_int.astype(np.float32)) * self.s["qact_pos"]
        c["dy_x"], c["dy_pos"] = _Dy(s["patch_embed.qact"], s["qact1"]), _Dy(s["qact_pos"], s["qact1"])
        self.blocks = []
        s_x = s["qact1"]
        for i in range(cfg.depth):
            p = f"blocks.{i}."
            b = {}
            bi, sc = orc.layernorm_consts(weights[p + "norm1.weight"], weights[p + "norm1.bias"], D)
            b["ln1"], b["dy_ln1"] = (_t(bi), _t(sc)), _Dy(sc, s[p + "qact1"])
            b["qkv"] = lin(p + "attn.qkv", s[p + "qact1"])
            b["dy_qkv"] = _Dy(b["qkv"][3], s[p + "attn.qact1"])
            s1 = s[p + "attn.qact1"]
            s_qk = np.float32(np.float32(s1 * s1) * np.float32(self.hs))
            b["s_qk"], b["dy_qk"] = torch.tensor(float(s_qk)), _Dy(s_qk, s[p + "attn.qact_attn1"])
            s_av = np.float32(np.float32(2.0 ** -15) * s1)
            b["dy_av"] = _Dy(s_av, s[p + "attn.qact2"])
            b["proj"] = lin(p + "attn.proj", s[p + "attn.qact2"])
            b["dy_proj"] = _Dy(b["proj"][3], s[p + "attn.qact3"])
            b["dy_res1"] = (_Dy(s[p + "attn.qact3"], s[p + "qact2"]), _Dy(s_x, s[p + "qact2"]))
            s_x = s[p + "qact2"]
            bi, sc = orc.layernorm_consts(weights[p + "norm2.weight"], weights[p + "norm2.bias"], D)
            b["ln2"], b["dy_ln2"] = (_t(bi), _t(sc)), _Dy(sc, s[p + "qact3"])
            b["fc1"] = lin(p + "mlp.fc1", s[p + "qact3"])
            b["dy_fc1"] = _Dy(b["fc1"][3], s[p + "mlp.qact_gelu"])
            b["dy_gelu"] = _Dy(np.float32(s[p + "mlp.qact_gelu"] * np.float32(2.0 ** -7)), s[p + "mlp.qact1"])
            b["fc2"] = lin(p + "mlp.fc2", s[p + "mlp.qact1"])
            b["dy_fc2"] = _Dy(b["fc2"][3], s[p + "mlp.qact2"])
            b["dy_res2"] = (_Dy(s[p + "mlp.qact2"], s[p + "qact4"]), _Dy(s_x, s[p + "qact4"]))
            s_x = s[p + "qact4"]
            self.blocks.append(b)
        bi, sc = orc.layernorm_consts(weights["norm.weight"], weights["norm.bias"], D)
        c["ln"], c["dy_ln"] = (_t(bi), _t(sc)), _Dy(sc, s["qact2"])
        c["head"] = lin("head", s["qact2"])
        self.c = c

    @torch.no_grad()
    def forward(self, images_int8):
        """int8 images [B, 3, H, W] (numpy or tensor) -> (fp32 logits, per-class scale); logits / scale = int32 accumulators"""
        cfg, c, s = self.cfg, self.c, self.s
        x = _t(np.asarray(images_int8), torch.float32) * s["qact_input"]
        B = x.shape[0]
        P, D, H, dh, N = cfg.patch_size, cfg.embed_dim, cfg.num_heads, cfg.head_dim, cfg.num_tokens
        # PatchEmbed: the conv with kernel = stride = P is a linear map of the (c, ph, pw) patches
        g = cfg.img_size // P
        patches = x.reshape(B, cfg.in_chans, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, -1)
        x = quant_linear(patches, s["qact_input"], c["pe"][0], c["pe"][1], c["pe"][2])
        x = quant_act(x, c["pe_dy"], s["patch_embed.qact"], 16)
        x = torch.cat((c["cls"].expand(B, -1, -1), x), dim=1)
        x = quant_act(x, c["dy_x"], s["qact1"], 16, c["pos"].unsqueeze(0).expand(B, -1, -1), c["dy_pos"])
        s_x = s["qact1"]
        for i, b in enumerate(self.blocks):
            p = f"blocks.{i}."
            y, sc = int_layernorm(x, s_x, *b["ln1"])
            y = quant_act(y, b["dy_ln1"], s[p + "qact1"], 8)
            y = quant_linear(y, s[p + "qact1"], b["qkv"][0], b["qkv"][1], b["qkv"][2])
            y = quant_act(y, b["dy_qkv"], s[p + "attn.qact1"], 8)
            s1 = s[p + "attn.qact1"]
            qkv = y.reshape(B, N, 3, H, dh).permute(2, 0, 3, 1, 4)
            q, k, v = qkv[0], qkv[1], qkv[2]
            attn = ((q / s1) @ (k.transpose(-2, -1) / s1)) * (s1 * s1)
            attn = attn * self.hs
            attn = quant_act(attn, b["dy_qk"], s[p + "attn.qact_attn1"], 8)
            attn, s_p = int_softmax(attn, s[p + "attn.qact_attn1"], 16)
            y = ((attn / s_p) @ (v / s1)) * (s_p * s1)
            y = y.transpose(1, 2).reshape(B, N, D)
            y = quant_act(y, b["dy_av"], s[p + "attn.qact2"], 8)
            y = quant_linear(y, s[p + "attn.qact2"], b["proj"][0], b["proj"][1], b["proj"][2])
            y = quant_act(y, b["dy_proj"], s[p + "attn.qact3"], 16)
            x = quant_act(y, b["dy_res1"][0], s[p + "qact2"], 16, x, b["dy_res1"][1])
            s_x = s[p + "qact2"]
            y, sc = int_layernorm(x, s_x, *b["ln2"])
            y = quant_act(y, b["dy_ln2"], s[p + "qact3"], 8)
            y = quant_linear(y, s[p + "qact3"], b["fc1"][0], b["fc1"][1], b["fc1"][2])
            y = quant_act(y, b["dy_fc1"], s[p + "mlp.qact_gelu"], 8)
            y, s_g = int_gelu(y, s[p + "mlp.qact_gelu"])
            y = quant_act(y, b["dy_gelu"], s[p + "mlp.qact1"], 8)
            y = quant_linear(y, s[p + "mlp.qact1"], b["fc2"][0], b["fc2"][1], b["fc2"][2])
            y = quant_act(y, b["dy_fc2"], s[p + "mlp.qact2"], 16)
            x = quant_act(y, b["dy_res2"][0], s[p + "qact4"], 16, x, b["dy_res2"][1])
            s_x = s[p + "qact4"]
        y, sc = int_layernorm(x, s_x, *c["ln"])
        y = y[:, 0]
        y = quant_act(y, c["dy_ln"], s["qact2"], 8)
        logits = quant_linear(y, s["qact2"], c["head"][0], c["head"][1], c["head"][2])
        return logits, c["head"][2]
