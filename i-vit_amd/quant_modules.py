"""Drop-in operator surface: the seven classes of the reference
`models/quantization_utils/quant_modules.py` (same names, constructor arguments, error
behaviour and `forward(x, scale...) -> (y, scale)` shape), running on MI355X through the
C-ABI of include/ivit.h.

Two tensor conventions, chosen per call by what the caller passes (SURVEY.md §8b "accepting either"):

  * INTEGER (native, what the fused engines and this package's own models use): activations are
    integer device tensors (int8 / int16 / int32; IntLayerNorm's integer-valued fp32 `IntValued`)
    plus an fp32 scale held on the host — the kernels re-derive fl(Q*s) where its rounding matters
    (SURVEY.md Appendix A).
  * FAKE-QUANT fp32 (the reference's own convention, quant_modules.py:204-206): `X = fl(Q*s)` with
    its scale.  A floating-point activation is converted ONCE per operator with the reference's own
    `rne(fl(X / s))` (quant_utils.py:220; :94, :359, :426, :484 consume the same quotient), the
    integer kernel runs, and the result goes back as `fl(Q_out * s_out)` — so reference caller code
    (e.g. Attention.forward, vit_quant.py:59-88, including its `attn * self.scale`) runs unchanged.
    A tensor that is not on the integer grid of its scale (a float mask added to logits, say) is
    refused with a ValueError instead of being rounded silently.

The frozen ("fixed") inference path and the `running_stat=True` calibration branch of QuantAct
(quant_modules.py:170-192) are implemented.
There is no CPU fallback: without the HIP library / a GPU every forward raises.
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from . import freeze as fz

_P = ctypes.c_void_p
_handles = {}


def handle(device):
    """one ivit_handle per device, bound to torch's current stream at each call"""
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.IvitError("ivit_amd operators run on a HIP device only (no CPU fallback)")
    idx = device.index if device.index is not None else torch.cuda.current_device()
    h = _handles.get(idx)
    if h is None:
        h = _handles[idx] = _lib.Handle(idx, torch.cuda.current_stream(idx).cuda_stream)
    h.set_stream(torch.cuda.current_stream(idx).cuda_stream)
    return h


def _ptr(t):
    return _P(t.data_ptr())


def _f32(s):
    """scale -> numpy float32 array (host)"""
    if isinstance(s, torch.Tensor):
        s = s.detach().cpu().numpy()
    return np.atleast_1d(np.asarray(s, np.float32)).reshape(-1)


def _dyv(d):
    return _lib.Dyadic(float(d[0, 0]), float(d[0, 1]))


def _leading_broadcast(id_shape, x_shape):
    """True if `identity` broadcasts to x only over leading dims (trailing dims identical)."""
    id_shape = tuple(id_shape)
    while id_shape and id_shape[0] == 1:
        id_shape = id_shape[1:]
    return len(id_shape) <= len(x_shape) and tuple(x_shape[len(x_shape) - len(id_shape):]) == id_shape


def to_int(x_fp32, scale, dtype=torch.int32):
    """integer view of a reference-style fake-quant tensor: rne(x / s) (quant_utils.py:220)"""
    s = torch.as_tensor(_f32(scale), device=x_fp32.device)
    return torch.round(x_fp32 / s).to(dtype)


class IntValued(torch.Tensor):
    """fp32 tensor whose VALUES are integers (IntLayerNorm's output in the integer convention: it can
    exceed int32).  The subclass is the marker that tells QuantAct not to treat it as fake-quant."""


def _is_fake(x):
    return x.is_floating_point() and not isinstance(x, IntValued)


def _scale_t(scale, x, channel_dim=-1):
    """scale (scalar or per-channel) as an fp32 device tensor broadcastable against x"""
    sv = _f32(scale)
    t = torch.as_tensor(sv, device=x.device)
    if sv.size == 1:
        return t.reshape(())
    shape = [1] * x.dim()
    shape[channel_dim] = sv.size
    return t.reshape(shape)


def from_fake(x, scale, dtype, lo, hi, what, channel_dim=-1):
    """fake-quant fp32 X = fl(Q*s) -> Q with the reference's rne(fl(X / s)); refuses off-grid or out-of-range input"""
    quo = x.float() / _scale_t(scale, x, channel_dim)
    q = torch.round(quo)
    bad = (q - quo).abs().max().item() if x.numel() else 0.0
    if bad > 0.05:
        raise ValueError(f"{what}: input is not a fake-quant tensor of the given scale (|x/s - rne(x/s)| up to {bad:.3f}); "
                         "non-integer products must be folded into the scale or passed separately (e.g. IntSoftmax mask=)")
    if q.numel() and (q.min().item() < lo or q.max().item() > hi):
        raise ValueError(f"{what}: rne(x/s) leaves [{lo}, {hi}] — wrong scale or wrong bit width")
    return q.to(dtype)


def to_fake(q, scale, channel_dim=-1):
    """Q, s -> fl(Q*s) fp32 (quant_modules.py:204-206)"""
    return q.float().as_subclass(torch.Tensor) * _scale_t(scale, q, channel_dim)


class _DyCache:
    """device dyadic tables keyed by the (pre-scale bytes, own scale) pair — the reference
    recomputes them with a host Decimal loop on every call (quant_utils.py:150-175)."""

    def __init__(self):
        self.c = {}

    def get(self, s_pre, s_out, device):
        key = (s_pre.tobytes(), np.float32(s_out).tobytes(), str(device))
        v = self.c.get(key)
        if v is None:
            d = fz.dyadic(s_pre, s_out)
            v = self.c[key] = (d, torch.from_numpy(d).to(device))
        return v


class QuantLinear(nn.Linear):
    """reference quant_modules.py:12-97"""

    def __init__(self, in_features, out_features, bias=True, weight_bit=8, bias_bit=32, per_channel=True,
                 quant_mode="symmetric"):
        super().__init__(in_features, out_features, bias)
        self.weight_bit, self.per_channel, self.bias_bit = weight_bit, per_channel, bias_bit
        self.quantize_bias = bias_bit is not None
        self.quant_mode = quant_mode
        if quant_mode == "asymmetric":
            raise NotImplementedError("unsupported quant mode: {}".format(quant_mode))
        if quant_mode != "symmetric":
            raise ValueError("unknown quant mode: {}".format(quant_mode))
        if weight_bit != 8 or bias_bit != 32:
            raise NotImplementedError("the MI355X path implements 8-bit weights / 32-bit bias")
        self.register_buffer("fc_scaling_factor", torch.zeros(out_features))
        self.register_buffer("weight_integer", torch.zeros_like(self.weight))
        if self.bias is not None:
            self.register_buffer("bias_integer", torch.zeros_like(self.bias))
        self._frozen = None

    def fix(self):
        pass

    def unfix(self):
        pass

    def _prepare(self, s_x, device):
        # the frozen integers depend on the float parameters: their in-place version counters are part of the key, so
        # load_state_dict / copy_ after a forward re-quantises instead of silently keeping the old int8 weights
        key = (np.float32(s_x).tobytes(), str(device), self.weight._version, id(self.weight),
               None if self.bias is None else (self.bias._version, id(self.bias)))
        if self._frozen is not None and self._frozen[0] == key:
            return self._frozen[1]
        if not self.per_channel:
            raise Exception("For weight, we only support per_channel quantization.")
        w = self.weight.detach().cpu().numpy()
        w_int, s_w = fz.quantize_weight(w)
        if self.bias is not None:
            b_int, s_b = fz.quantize_bias(self.bias.detach().cpu().numpy(), s_w, s_x)
        else:
            b_int, s_b = None, (s_w * np.float32(s_x)).astype(np.float32)
        c = dict(w=torch.from_numpy(w_int).to(device),
                 b=None if b_int is None else torch.from_numpy(b_int).to(device), s_b=s_b)
        self.fc_scaling_factor = torch.from_numpy(s_w)
        self.weight_integer = torch.from_numpy(w_int.astype(np.float32))
        if b_int is not None:
            self.bias_integer = torch.from_numpy(b_int.astype(np.float32))
        self._frozen = (key, c)
        return c

    def forward(self, x, prev_act_scaling_factor=None):
        """x: int8 device tensor [..., in]; returns (int32 accumulators [..., out], scale[out])"""
        s_x = _f32(prev_act_scaling_factor)
        if s_x.size != 1:
            raise ValueError("QuantLinear expects a per-tensor input scale")
        fake = _is_fake(x)
        if fake:
            x = from_fake(x, s_x, torch.int8, -128, 127, "QuantLinear")
        if x.dtype != torch.int8:
            raise TypeError("QuantLinear input must be int8 or a fake-quant fp32 tensor (use QuantAct first)")
        c = self._prepare(s_x[0], x.device)
        x2 = x.reshape(-1, self.in_features).contiguous()
        acc = torch.empty(x2.shape[0], self.out_features, dtype=torch.int32, device=x.device)
        handle(x.device).call("ivit_linear_i8", _ptr(x2), _ptr(c["w"]), _ptr(c["b"]) if c["b"] is not None else None,
                              _ptr(acc), x2.shape[0], self.out_features, self.in_features)
        acc = acc.reshape(*x.shape[:-1], self.out_features)
        s_b = torch.from_numpy(c["s_b"])
        return (to_fake(acc, s_b) if fake else acc), s_b


class QuantAct(nn.Module):
    """reference quant_modules.py:100-206 (frozen path: fixedpoint_mul, quant_utils.py:192-253)"""

    def __init__(self, activation_bit=8, act_range_momentum=0.95, running_stat=True, per_channel=False,
                 quant_mode="symmetric"):
        super().__init__()
        self.activation_bit, self.act_range_momentum = activation_bit, act_range_momentum
        self.running_stat, self.quant_mode, self.per_channel = running_stat, quant_mode, per_channel
        self.min_val = torch.zeros(1)
        self.max_val = torch.zeros(1)
        self.register_buffer("act_scaling_factor", torch.zeros(1))
        if quant_mode == "asymmetric":
            raise NotImplementedError("unsupported quant mode: {}".format(quant_mode))
        if quant_mode != "symmetric":
            raise ValueError("unknown quant mode: {}".format(quant_mode))
        self._dy = _DyCache()
        # reference callers expect the input QuantAct (no pre-scale) to hand back fl(q*s) fp32; the integer convention
        # (default) returns the int8 tensor itself
        self.fake_quant_input = False

    def fix(self):
        self.running_stat = False

    def unfix(self):
        self.running_stat = True

    def set_scale(self, s):
        self.act_scaling_factor = torch.tensor([float(np.float32(s))], dtype=torch.float32)

    def set_range(self, min_val, max_val):
        """scale from a calibrated range exactly like quant_utils.py:51-69"""
        self.set_scale(fz.symmetric_scale(min_val, max_val, self.activation_bit))

    @torch.no_grad()
    def quantize_param(self, param):
        """input branch applied to a float PARAMETER on the host (pos_embed through qact_pos,
        vit_quant.py:264): calibrates from the parameter when running_stat, returns float64 integers."""
        p = param.detach().cpu().float()
        if self.running_stat:
            self._collect_range(p, None, None, None)
        s = np.float32(self.act_scaling_factor.reshape(-1)[0].item())
        if not s > 0:
            raise ValueError("QuantAct has no scale: load act_scaling_factor or calibrate first")
        return fz.quantize(p.numpy(), s, self.activation_bit, False)

    @torch.no_grad()
    def _collect_range(self, x, s_pre, identity, s_id):
        """calibration (reference quant_modules.py:170-192): track min/max of the fp32 activation this
        QuantAct sees — fl(x_int * s_pre) (+ fl(id_int * s_id)) — with the reference's momentum rule,
        then its scale (quant_utils.py:51-69).  Range statistics are torch reductions on the device;
        they run in calibration only, never on the frozen inference path."""
        ref_val = getattr(x, "_calib_fp32", None)      # IntGELU in calibration mode: the reference's own fp32 tensor
        if ref_val is not None and identity is None:     # both conventions: the integer pair and the fake-quant tensor
            X = ref_val
        elif s_pre is None:
            X = x.float()
        else:
            X = x.float() * torch.as_tensor(_f32(s_pre), device=x.device)
            if identity is not None:
                X = identity.float() * torch.as_tensor(_f32(s_id), device=x.device) + X
        cur_min = np.float32(X.min().item())
        cur_max = np.float32(X.max().item())
        mn, mx = np.float32(self.min_val.reshape(-1)[0].item()), np.float32(self.max_val.reshape(-1)[0].item())
        if mn == mx:
            mn, mx = cur_min, cur_max
        else:
            mom = np.float32(self.act_range_momentum)
            mn = np.float32(np.float32(mn * mom) + np.float32(cur_min * np.float32(1 - self.act_range_momentum)))
            mx = np.float32(np.float32(mx * mom) + np.float32(cur_max * np.float32(1 - self.act_range_momentum)))
        self.min_val, self.max_val = torch.tensor([mn]), torch.tensor([mx])
        self.set_range(mn, mx)

    def forward(self, x, pre_act_scaling_factor=None, identity=None, identity_scaling_factor=None):
        fake = pre_act_scaling_factor is not None and _is_fake(x)
        if self.running_stat:
            if fake or pre_act_scaling_factor is None:
                # the fp32 activation itself (reference quant_modules.py:167: x_act = x if identity is None else identity + x)
                self._collect_range(x if identity is None else identity.float() + x.float(), None, None, None)
            else:
                self._collect_range(x, pre_act_scaling_factor, identity, identity_scaling_factor)
        s_out = np.float32(self.act_scaling_factor.reshape(-1)[0].item())
        if not s_out > 0:
            raise ValueError("QuantAct has no scale: load act_scaling_factor or call set_scale()")
        bits = self.activation_bit
        h = handle(x.device)
        out_dt = {8: torch.int8, 16: torch.int16, 32: torch.int32}.get(bits)
        if out_dt is None:
            raise NotImplementedError("activation_bit must be 8, 16 or 32 on the MI355X path")
        if pre_act_scaling_factor is None:
            if bits != 8 or not x.is_floating_point():
                raise NotImplementedError("input quantisation is implemented for fp32 -> 8 bit")
            xc = x.contiguous().float()
            q = torch.empty(xc.shape, dtype=torch.int8, device=x.device)
            h.call("ivit_quantize_input_f32", _ptr(xc), float(s_out), _ptr(q), xc.numel())
            if self.fake_quant_input:
                return to_fake(q, self.act_scaling_factor), self.act_scaling_factor
            return q, self.act_scaling_factor
        s_pre = _f32(pre_act_scaling_factor)
        # channel dim: last, or dim 1 for the conv layout [B, C, H, W] with a (1, C, 1, 1) scale (quant_modules.py:316-330)
        ps = tuple(pre_act_scaling_factor.shape) if isinstance(pre_act_scaling_factor, torch.Tensor) else ()
        conv_layout = len(ps) == 4 and x.dim() == 4 and ps[1] == x.shape[1] and ps[1] > 1
        cdim = 1 if conv_layout else -1
        if fake:
            # the reference's own first step, z = rne(fl(X / s_pre)) in fp32 (quant_utils.py:220): it can exceed int32 after
            # I-LayerNorm, so it stays an integer-valued fp32 tensor for the fp64 requant kernel
            quo = x.float() / _scale_t(s_pre, x, cdim)
            # A plain fp32 tensor is taken as the reference's fake-quant X = Q * s_pre.  Integers are passed as an int dtype or
            # as IntValued (the explicit convention; torch ops inside __torch_function__ keep the marker).  Only the gross
            # case of a lost marker is caught below — an I-LayerNorm-sized integer tensor; a small integer-valued fp32
            # tensor without the marker is indistinguishable from fake-quant values on a coarse grid and IS divided by the
            # scale (documented limit).  An integer-valued tensor that lost
            # its IntValued marker on the way (.numpy() / .data / an op outside __torch_function__) would be divided by the
            # scale a second time: quotients no operator of the path can produce (the I-LayerNorm output, the largest, stays
            # below 2^40) together with all-integer values are that case — refuse instead of returning other numbers
            if quo.numel() and quo.abs().max().item() >= 2.0 ** 44 and bool((x.float() == torch.round(x.float())).all()):
                raise ValueError("QuantAct: integer-valued fp32 input without the IntValued marker (x / s_pre reaches "
                                 f"{quo.abs().max().item():.3g}); pass integers as an int dtype or re-wrap with IntValued")
            x = torch.round(quo).as_subclass(IntValued)
            if identity is not None and _is_fake(identity):     # an integer identity (e.g. a quantised parameter table) stays
                identity = from_fake(identity, identity_scaling_factor, torch.int32, -2 ** 31, 2 ** 31 - 1, "QuantAct identity")
        if conv_layout:
            x = x.permute(0, 2, 3, 1).contiguous()
            if x.is_floating_point():
                x = x.as_subclass(IntValued)
            if identity is not None:
                identity = identity.permute(0, 2, 3, 1).contiguous()
        out = self._requant(x, s_pre, s_out, identity, identity_scaling_factor, bits, out_dt, h)
        if conv_layout:
            out = out.permute(0, 3, 1, 2)
        if fake:
            return to_fake(out, self.act_scaling_factor), self.act_scaling_factor
        return out, self.act_scaling_factor

    def _requant(self, x, s_pre, s_out, identity, identity_scaling_factor, bits, out_dt, h):
        """integer convention: x int32 / int16 / int8 or integer-valued fp32, channel = last dim"""
        C = x.shape[-1]
        if s_pre.size not in (1, C):
            raise NotImplementedError("scale must be per-tensor or per-channel on the last dim")
        d, dd = self._dy.get(s_pre, s_out, x.device)
        zi = di = None
        if identity is not None:
            s_id = _f32(identity_scaling_factor)
            if s_id.size != 1:
                raise NotImplementedError("identity scale must be per-tensor")
            di_host, di = self._dy.get(s_id, s_out, x.device)
            if (identity.numel() < x.numel() and s_pre.size == 1 and x.dtype != torch.float32
                    and bits in (8, 16) and _leading_broadcast(identity.shape, x.shape)):
                zc = x.to(torch.int32).contiguous()
                idc = identity.to(torch.int32).contiguous()
                out = torch.empty(x.shape, dtype=out_dt, device=x.device)
                h.call("ivit_requant_i32_bcast", _ptr(zc), _dyv(d), _ptr(idc), idc.numel(), _dyv(di_host), bits,
                       _ptr(out), zc.numel())
                return out
            zi = identity.to(torch.int32).expand(x.shape).contiguous()
        out = torch.empty(x.shape, dtype=out_dt, device=x.device)
        rows = x.numel() // C
        if x.dtype == torch.float32:
            xc = x.as_subclass(torch.Tensor).contiguous()
            h.call("ivit_requant_f32", _ptr(xc), _ptr(dd), d.shape[0], _ptr(zi) if zi is not None else None,
                   _ptr(di) if di is not None else None, bits, _ptr(out), rows, C)
        else:
            xc = x.to(torch.int32).contiguous()
            h.call("ivit_requant_i32", _ptr(xc), _ptr(dd), d.shape[0], _ptr(zi) if zi is not None else None,
                   _ptr(di) if di is not None else None, bits, _ptr(out), rows, C)
        return out


class QuantMatMul(nn.Module):
    """reference quant_modules.py:209-228; A int8 (or 16-bit Shiftmax output), B int8"""

    def __init__(self):
        super().__init__()
        self.register_buffer("act_scaling_factor", torch.zeros(1))

    def fix(self):
        pass

    def unfix(self):
        pass

    def forward(self, A, pre_act_scaling_factor_A, B, pre_act_scaling_factor_B):
        sA, sB = _f32(pre_act_scaling_factor_A), _f32(pre_act_scaling_factor_B)
        s_out = (sA * sB).astype(np.float32)
        self.act_scaling_factor = torch.from_numpy(s_out)
        fake = _is_fake(A) or _is_fake(B)
        if _is_fake(A):      # int8 operand, or the 16-bit Shiftmax output (0 .. 32768)
            qa = torch.round(A.float() / _scale_t(sA, A))
            wide = bool(qa.numel() and (qa.max().item() > 127 or qa.min().item() < -128))
            A = from_fake(A, sA, torch.int32 if wide else torch.int8, 0 if wide else -128, 32768 if wide else 127, "QuantMatMul A")
        if _is_fake(B):
            B = from_fake(B, sB, torch.int8, -128, 127, "QuantMatMul B")
        if B.dtype != torch.int8:
            raise TypeError("QuantMatMul: B must be int8")
        M, K = A.shape[-2], A.shape[-1]
        N = B.shape[-1]
        batch = A.shape[:-2]
        nb = int(np.prod(batch)) if len(batch) else 1
        Kp = (K + 15) // 16 * 16
        # "NT" operand: B^T with K contiguous, row stride padded to 16 (layout plumbing)
        Bt = torch.zeros(nb, N, Kp, dtype=torch.int8, device=A.device)
        Bt[:, :, :K] = B.reshape(nb, K, N).transpose(1, 2)
        C = torch.empty(nb, M, N, dtype=torch.int32, device=A.device)
        h = handle(A.device)
        if A.dtype == torch.int8:
            Ap = torch.zeros(nb, M, Kp, dtype=torch.int8, device=A.device)
            Ap[:, :, :K] = A.reshape(nb, M, K)
            h.call("ivit_bmm_nt_i8", _ptr(Ap), _ptr(Bt), _ptr(C), nb, M, N, K, Kp, Kp, N, M * Kp, N * Kp, M * N)
        else:
            Ap = torch.zeros(nb, M, Kp, dtype=torch.int16, device=A.device)   # uint16 payload
            Ap[:, :, :K] = A.reshape(nb, M, K).to(torch.int32).to(torch.int16)
            h.call("ivit_bmm_nt_u16i8", _ptr(Ap), _ptr(Bt), _ptr(C), nb, M, N, K, Kp, Kp, N, M * Kp, N * Kp, M * N)
        C = C.reshape(*batch, M, N)
        return (to_fake(C, self.act_scaling_factor) if fake else C), self.act_scaling_factor


class QuantConv2d(nn.Conv2d):
    """reference quant_modules.py:231-330; the patch-embedding case kernel == stride, no padding"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, weight_bit=8, bias_bit=32, quant_mode="symmetric", per_channel=True,
                 weight_percentile=0):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        self.weight_bit, self.quant_mode, self.per_channel = weight_bit, quant_mode, per_channel
        self.weight_percentile, self.bias_bit = weight_percentile, bias_bit
        self.quantize_bias = bias_bit is not None
        self.register_buffer("conv_scaling_factor", torch.zeros(out_channels))
        self.register_buffer("weight_integer", torch.zeros_like(self.weight))
        self.register_buffer("bias_integer", torch.zeros_like(self.bias))
        self._frozen = None

    def fix(self):
        pass

    def unfix(self):
        pass

    def forward(self, x, pre_act_scaling_factor=None):
        if self.quant_mode == "asymmetric":
            raise NotImplementedError("unsupported quant mode: {}".format(self.quant_mode))
        if self.quant_mode != "symmetric":
            raise ValueError("unknown quant mode: {}".format(self.quant_mode))
        if not self.per_channel:
            raise Exception("For weight, we only support per_channel quantization.")
        P = self.kernel_size[0]
        if (self.kernel_size != self.stride or self.kernel_size[0] != self.kernel_size[1] or
                any(self.padding) or self.groups != 1 or any(d != 1 for d in self.dilation)):
            raise NotImplementedError("QuantConv2d on MI355X: non-overlapping patch convolution only")
        s_x = _f32(pre_act_scaling_factor)[0]
        fake = _is_fake(x)
        if fake:
            x = from_fake(x, s_x, torch.int8, -128, 127, "QuantConv2d")
        key = (np.float32(s_x).tobytes(), str(x.device), self.weight._version, id(self.weight), self.bias._version, id(self.bias))
        if self._frozen is None or self._frozen[0] != key:
            w_int, s_w = fz.quantize_weight(self.weight.detach().cpu().numpy())
            b_int, s_b = fz.quantize_bias(self.bias.detach().cpu().numpy(), s_w, s_x)
            self.conv_scaling_factor = torch.from_numpy(s_w)
            self._frozen = (key, dict(w=torch.from_numpy(w_int.reshape(w_int.shape[0], -1)).to(x.device),
                                      b=torch.from_numpy(b_int).to(x.device), s_b=s_b))
        c = self._frozen[1]
        B, Cin, Hh, Ww = x.shape
        K = Cin * P * P
        rows = torch.empty(B * (Hh // P) * (Ww // P), K, dtype=torch.int8, device=x.device)
        h = handle(x.device)
        xc = x.contiguous()
        h.call("ivit_im2col_patch", _ptr(xc), B, Cin, Hh, Ww, P, _ptr(rows))
        acc = torch.empty(rows.shape[0], self.out_channels, dtype=torch.int32, device=x.device)
        h.call("ivit_linear_i8", _ptr(rows), _ptr(c["w"]), _ptr(c["b"]), _ptr(acc), rows.shape[0],
               self.out_channels, K)
        y = acc.reshape(B, Hh // P, Ww // P, self.out_channels).permute(0, 3, 1, 2)
        s_b = torch.from_numpy(c["s_b"]).view(1, -1, 1, 1)
        return (to_fake(y, s_b, channel_dim=1) if fake else y), s_b


class IntLayerNorm(nn.LayerNorm):
    """reference quant_modules.py:333-386 (I-LayerNorm)"""

    def __init__(self, normalized_shape, eps=1e-5, elementwise_affine=True):
        super().__init__(normalized_shape, eps, elementwise_affine)
        self.dim_sqrt = None
        self.register_buffer("norm_scaling_factor", torch.zeros(1))
        self.register_buffer("bias_integer", torch.zeros_like(self.bias))
        self._frozen = None
        # The reference's two row sums follow torch's summation order, which depends on the MEMORY
        # LAYOUT of its fp32 activation: "channel" (contiguous last dim, the usual case) or "token"
        # (Swin stage 0, where activations keep the layout of flatten(2).transpose(1,2)).
        self.sum_order = "channel"

    def fix(self):
        pass

    def unfix(self):
        pass

    def forward(self, x, scaling_factor=None):
        """x int16 [B, N, C]; returns (z float32 integer-valued [B,N,C], scale[C]) — z is the
        integer the following QuantAct derives (it can exceed int32)."""
        s = _f32(scaling_factor)[0]
        fake = _is_fake(x)
        if fake:
            x = from_fake(x, s, torch.int16, -32768, 32767, "IntLayerNorm")
        key = (str(x.device), self.weight._version, id(self.weight), self.bias._version, id(self.bias))
        if self._frozen is None or self._frozen[0] != key:
            bi, sc = fz.layernorm_constants(self.weight.detach().cpu().numpy(), self.bias.detach().cpu().numpy())
            self._frozen = (key, torch.from_numpy(bi).to(x.device), torch.from_numpy(sc).to(x.device), sc)
            self.bias_integer = torch.from_numpy(bi)
            self.norm_scaling_factor = torch.from_numpy(sc)
        _, bi_d, sc_d, sc = self._frozen
        C = x.shape[-1]
        xc = x.to(torch.int16).contiguous()
        z = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        if self.sum_order == "token":
            handle(x.device).call("ivit_layernorm_tokenorder", _ptr(xc), xc.numel() // C, C, float(s), _ptr(bi_d),
                                  _ptr(sc_d), int(x.shape[-2]), _ptr(z))
        else:
            handle(x.device).call("ivit_layernorm", _ptr(xc), xc.numel() // C, C, float(s), _ptr(bi_d), _ptr(sc_d),
                                  _ptr(z))
        if fake:
            return to_fake(z, sc), torch.from_numpy(sc)
        return z.as_subclass(IntValued), torch.from_numpy(sc)


class IntGELU(nn.Module):
    """reference quant_modules.py:389-445 (ShiftGELU)"""

    def __init__(self, output_bit=8):
        super().__init__()
        self.output_bit = output_bit
        self.n = 23
        self.register_buffer("act_scaling_factor", torch.zeros(1))
        if output_bit != 8:
            raise NotImplementedError("ShiftGELU on MI355X: 8-bit sigmoid (the reference default)")
        # calibration (the QuantActs start in running_stat mode, freeze_model calls fix()): while set, the output carries the
        # reference's own fp32 value for the range statistics of the QuantAct that follows (see forward)
        self.calibrating = True

    def fix(self):
        self.calibrating = False

    def unfix(self):
        self.calibrating = True

    def forward(self, x, scaling_factor=None):
        s = np.float32(_f32(scaling_factor)[0])
        fake = _is_fake(x)
        if fake:
            x = from_fake(x, s, torch.int8, -128, 127, "IntGELU")
        C = x.shape[-1]
        xc = x.contiguous()
        out = torch.empty(x.shape, dtype=torch.int16, device=x.device)
        handle(x.device).call("ivit_shiftgelu", _ptr(xc), xc.numel() // C, C, float(s), _ptr(out))
        s_out = np.float32(s * np.float32(1.0 / 2 ** (self.output_bit - 1)))
        self.act_scaling_factor = torch.tensor([float(s_out)])
        res = to_fake(out, self.act_scaling_factor) if fake else out
        if self.calibrating:
            # What the reference hands to the next QuantAct is fl(fl(x_int * sigmoid_int) * s_out) with the NON-integer
            # x_int = fl(fl(Q s) / s) (quant_modules.py:441-445), one ulp away from fl(Q sigmoid_int s_out) on some elements;
            # a QuantAct in running_stat mode tracks min / max of exactly that tensor (:170-192).  It is elementwise and
            # deterministic, so calibration reproduces it (torch fp32 on the device: IEEE multiply and divide);
            # sigmoid_int = out / Q is exact (out = Q sigmoid_int, |out| < 2^15).  Calibration only.
            with torch.no_grad():
                qf = xc.float()
                st = torch.as_tensor(np.float32(s), device=x.device)
                pre = (qf * st) / st
                sig = torch.where(qf != 0, out.float() / torch.where(qf != 0, qf, torch.ones_like(qf)), torch.zeros_like(qf))
                res._calib_fp32 = (pre * sig) * torch.as_tensor(np.float32(s_out), device=x.device)
        return res, self.act_scaling_factor


class IntSoftmax(nn.Module):
    """reference quant_modules.py:448-497 (Shiftmax); output int32 tensor holding 0..2^(b-1)"""

    def __init__(self, output_bit=8):
        super().__init__()
        self.output_bit = output_bit
        self.n = 15
        self.register_buffer("act_scaling_factor", torch.zeros(1))
        if output_bit not in (8, 16):
            raise NotImplementedError("Shiftmax on MI355X: 8- or 16-bit output")

    def fix(self):
        pass

    def unfix(self):
        pass

    def forward(self, x, scaling_factor, mask=None, num_heads=1):
        """mask: optional float [nW, n, n] (0 / -100.0) — the reference adds it to the fp32 logits
        right before this module (swin_quant.py:151-156); with integer activations it is passed in.
        Fake-quant logits that already carry that mask are recognised by x < -50 where 128 s < 50 — i.e. the detection is
        tied to the reference's literal -100.0; another mask value has to be passed through `mask`.  (The x.min() test
        costs one device sync per call on the fake-quant path; the integer path has none.)"""
        s = np.float32(_f32(scaling_factor)[0])
        fake = _is_fake(x)
        if fake:
            if mask is None and 128.0 * float(s) < 50.0 and x.numel() and x.min().item() < -50.0:
                if x.dim() < 2 or x.shape[-2] != x.shape[-1]:
                    raise ValueError("IntSoftmax: fake-quant logits below -50 are taken for the reference's -100.0 shift mask "
                                     "(swin_quant.py:151-156) and must be [..., n, n]; got " + str(tuple(x.shape)))
                # The reference's Swin block adds its float mask to the fake-quant logits BEFORE this module
                # (swin_quant.py:151-156: attn + mask, mask in {0, -100.0}), so a caller running the reference's own model
                # code hands over X = fl(fl(Q*s) - 100) on the masked entries: off the grid of s, but an integer-domain
                # side input in disguise.  On-grid values stay within 128 s < 50, so x < -50 identifies the masked
                # entries; Q comes back from fl(x + 100) (the lost low bits are far below half a grid step), and the mask
                # goes to the kernel as the side input it is — one window per leading row block (nW = rows / n, H = 1).
                mk_full = torch.where(x < -50.0, torch.full_like(x, -100.0), torch.zeros_like(x)).float()
                x = from_fake((x.float() - mk_full), s, torch.int8, -128, 127, "IntSoftmax (masked logits)")
                mask, num_heads = mk_full.reshape(-1, x.shape[-1], x.shape[-1]), 1
            else:
                x = from_fake(x, s, torch.int8, -128, 127, "IntSoftmax")
        n = x.shape[-1]
        xc = x.contiguous()
        out = torch.empty(x.shape, dtype=torch.int16, device=x.device)    # uint16 payload
        if mask is not None:
            mk = mask.to(device=x.device, dtype=torch.float32).contiguous()
            handle(x.device).call("ivit_shiftmax_masked", _ptr(xc), xc.numel() // n, n, n, float(s), self.output_bit,
                                  _ptr(mk), int(mk.shape[0]), int(num_heads), _ptr(out), n)
        else:
            handle(x.device).call("ivit_shiftmax", _ptr(xc), xc.numel() // n, n, n, float(s), self.output_bit,
                                  _ptr(out), n)
        self.act_scaling_factor = torch.tensor([1.0 / 2 ** (self.output_bit - 1)])
        out = out.to(torch.int32) & 0xFFFF
        return (to_fake(out, self.act_scaling_factor) if fake else out), self.act_scaling_factor
