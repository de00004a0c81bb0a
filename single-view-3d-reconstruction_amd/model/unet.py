"""UNet depth regressors on MI355X: mirror of the reference's model/unet.py:15-118 (`Unet`) and :121-186 (`UNetMini`),
the stage in front of the hot path in BASELINE config 5.

Same constructor arguments, forward signature ((B,Cin,H,W) -> (B,Cout,H,W)) and parameter / buffer names, so `unet.*`
checkpoint entries load (trainer/trainer_scene_net.py:204-212).  The nn.Conv2d / nn.BatchNorm2d submodules only HOLD the
parameters.  backend="hip" (default on GPU tensors): every convolution block runs in the library's kernels as an IMPLICIT GEMM
(conv2d_igemm.hip: the A operand of the split-precision MFMA GEMM is gathered from the channels-last activation, LeakyReLU /
ReLU and the decoder's torch.cat applied on the way in; a decoder block writes its x2-upsampled input once; stride-2 backward-data
as four parity-class problems; the weight gradient through a gather loader in the dW kernel; the 1-channel output layer on
plain-FMA kernels), BatchNorm in the 3-D encoder's kernels on (B,1,H,W,C) views; activations are channels-last inside.  The explicit
patch-matrix path of rounds 2-3 (_ConvBlockFn) is kept for A/B (SVR_UNET_IGEMM=0) and for the bf16x3 / exact-f32 backward modes.
backend="stock": the layers as stock PyTorch-ROCm ops (MIOpen), kept for A/B measurements (15-50 ms per config-5 step)."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


PREPARE_MANY = os.environ.get("SVR_UNET_PREP_MANY", "1") != "0"    # "0": every block prepares its own planes (A/B)
SMALL_M = 1024      # output pixels (times batch) up to which a layer's forward product runs as a split-reduction GEMM


class _ConvBlockFn(torch.autograd.Function):
    """y (B,Ho,Wo,Cout) = conv_k( [upsample x2]( act( cat(src0, src1) ) ) ) + bias, channels-last."""

    @staticmethod
    def forward(ctx, src0, src1, weight, bias, k, stride, act, up):
        src0 = src0.contiguous()
        src1 = src1.contiguous() if src1 is not None else None
        Cout = weight.shape[0]
        col, (Ho, Wo) = ops.conv2d_im2col(src0, src1, k, stride, act, up)
        wp = weight.detach().permute(0, 2, 3, 1).reshape(Cout, -1)              # [Cout][(ky*k+kx)*C + c]
        npad = Cout if Cout % 32 == 0 else (Cout + 31) // 32 * 32                # the GEMMs want 32 | N (dconv8: Cout = 1)
        if npad != Cout:
            wp = torch.cat([wp, wp.new_zeros(npad - Cout, wp.shape[1])])
            b = torch.cat([bias.detach(), bias.new_zeros(npad - Cout)])
        else:
            b = bias.detach()
        wp = wp.contiguous()
        M, K = col.shape
        if M <= SMALL_M and M % 4 == 0 and K >= 1024:
            # deep layers (1x1 .. 16x16 pixels): Y = col W^T has a handful of output tiles and a reduction of 2 304 - 4 608,
            # i.e. 8 workgroups walking ~290 k-steps (0.19 ms each, latency bound).  Y^T = (W^T)^T col^T is the shape of the
            # weight-gradient GEMM: reduction over the ROWS, split over workgroups, slabs summed in a fixed order --
            # the exact-f32 MFMA kernel (the FLOPs are negligible here).
            yT, _ = ops.linear_bwd_weight(wp.t().contiguous(), col.t().contiguous(), want_bias=False, mode="f32")
            y = yT.t() + b
        else:
            y = ops.linear_fwd(col, wp, b, relu=False)
        ctx.save_for_backward(src0, src1, wp)
        ctx.cfg = (k, stride, act, up, Cout, npad, tuple(weight.shape))
        B = src0.shape[0]
        return y[:, :Cout].reshape(B, Ho, Wo, Cout) if npad != Cout else y.view(B, Ho, Wo, Cout)

    @staticmethod
    def backward(ctx, dy):
        src0, src1, wp = ctx.saved_tensors
        k, stride, act, up, Cout, npad, wshape = ctx.cfg
        dy2 = dy.reshape(-1, Cout)
        if npad != Cout:
            dyp = dy2.new_zeros(dy2.shape[0], npad)
            dyp[:, :Cout] = dy2
            dy2 = dyp
        dy2 = dy2.contiguous()
        col, _ = ops.conv2d_im2col(src0, src1, k, stride, act, up)              # recomputed: cheaper than keeping 1.2 GB
        dwp, db = ops.linear_bwd_weight(dy2, col)
        del col
        dw = dwp[:Cout].view(Cout, k, k, -1).permute(0, 3, 1, 2).contiguous().view(wshape)
        need0, need1 = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and src1 is not None
        d0 = d1 = None
        if need0 or need1:
            dcol = ops.linear_bwd_data(dy2, wp)
            d0, d1 = ops.conv2d_col2im(src0, src1, k, stride, act, up, dcol, need0, need1)
        return d0, d1, dw, db[:Cout].contiguous(), None, None, None, None


class _ConvBlockIgemmFn(torch.autograd.Function):
    """The same block as implicit GEMMs (conv2d_igemm.hip): no patch matrix, the weight gradient comes out in nn.Conv2d's own
    layout, no torch glue ops.  A decoder block writes its activated, upsampled, concatenated input V once and keeps it for
    the backward (1x the activation; the patch matrix was 9x and was built twice)."""

    @staticmethod
    def forward(ctx, src0, src1, weight, bias, k, stride, act, up, *rest):
        planes = rest[0] if rest else None           # optional ninth argument: planes prepared by the caller
        ctx.nrest = len(rest)
        src0 = src0.contiguous()
        src1 = src1.contiguous() if src1 is not None else None
        want_bwd = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        if planes is None or (want_bwd and not planes.has_bwd):      # (the network prepares all its layers at once: _UNetBase.forward)
            planes = ops.Conv2dPlanes(weight, stride, want_bwd=want_bwd)
        if up:
            a0, a1, aact = ops.conv2d_virtual(src0, src1, act, True), None, ops.ACT_NONE
        else:
            a0, a1, aact = src0, src1, act
        y = ops.conv2d_fwd(a0, a1, k, stride, aact, planes, bias.detach())
        ctx.save_for_backward(src0, src1, a0 if up else None)
        ctx.planes = planes
        ctx.cfg = (k, stride, act, up, int(weight.shape[0]))
        return y

    @staticmethod
    def backward(ctx, dy):
        src0, src1, V = ctx.saved_tensors
        k, stride, act, up, Cout = ctx.cfg
        a0, a1, aact = (V, None, ops.ACT_NONE) if up else (src0, src1, act)
        dy = dy.contiguous()
        dw, db = ops.conv2d_bwd_weight(a0, a1, k, stride, aact, dy, Cout)
        need0, need1 = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and src1 is not None
        d0 = d1 = None
        if need0 or need1:
            dvirt = ops.conv2d_bwd_data(a0, a1, k, stride, ctx.planes, dy)
            d0, d1 = ops.conv2d_finish_bwd(src0, src1, act, up, dvirt, need0, need1)
        return (d0, d1, dw, db, None, None, None, None) + (None,) * ctx.nrest


def _igemm_on():
    return ops.UNET_IGEMM and ops.BACKWARD_GEMM == "f16x3s"


def _conv_block(*args, planes=None):
    """The block's autograd function: implicit GEMMs with the f32-level backward split (default), the explicit patch-matrix
    path for the other backward arithmetics (SVR_BACKWARD=bf16x3 / f32) and for SVR_UNET_IGEMM=0 (A/B)."""
    if _igemm_on():
        return _ConvBlockIgemmFn.apply(*args, planes)
    return _ConvBlockFn.apply(*args)


class _BN2dFn(torch.autograd.Function):
    """BatchNorm2d on a channels-last (B,H,W,C) tensor through the 3-D encoder's BatchNorm kernels (D = 1)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn):
        B, H, W, C_ = x.shape
        x5 = x.contiguous().view(B, 1, H, W, C_)
        y, _, _, ss, mean = ops.bn_forward(x5, gamma.detach(), beta.detach(), bn.running_mean, bn.running_var, bn.training,
                                           eps=bn.eps, momentum=bn.momentum, want_pool=False)
        if bn.training:
            bn.num_batches_tracked += 1
        ctx.save_for_backward(x5, ss, mean)
        ctx.training = bn.training
        return y.view(B, H, W, C_)

    @staticmethod
    def backward(ctx, dy):
        x5, ss, mean = ctx.saved_tensors
        dx, dgamma, dbeta = ops.bn_backward(x5, dy.contiguous().view(x5.shape), None, None, mean, ss, relu_mask=False,
                                            training=ctx.training)
        return dx.view(dy.shape), dgamma, dbeta, None


class _UNetBase(nn.Module):
    # encoder widths as multiples of num_filters; decoder: (name, in mult, out mult or None=channels_out, bn name)
    ENC = ()
    ENC_BN = ()
    DEC = ()

    def __init__(self, num_filters=32, channels_in=3, channels_out=3, backend="hip"):
        super().__init__()
        nf = num_filters
        c_prev = channels_in
        for i, mult in enumerate(self.ENC, start=1):
            setattr(self, f"conv{i}", nn.Conv2d(c_prev, nf * mult, 4, 2, 1))
            c_prev = nf * mult
        self.up = nn.Upsample(scale_factor=2, mode="bilinear")
        for name, cin, cout, _ in self.DEC:
            setattr(self, name, nn.Conv2d(nf * cin, channels_out if cout is None else nf * cout, 3, 1, 1))
        for name, mult in self.BN:
            setattr(self, name, nn.BatchNorm2d(nf * mult))
        self.leaky_relu = nn.LeakyReLU(0.2)
        self.relu = nn.ReLU()
        self.backend = backend

    def forward(self, input):
        if self.backend == "stock":
            return self._forward_stock(input)
        if not input.is_cuda:
            raise RuntimeError("UNet HIP path needs GPU tensors (no CPU fallback; backend='stock' runs stock torch ops)")
        x = input.float().permute(0, 2, 3, 1).contiguous()                       # NCHW -> channels-last
        planes = {}
        if _igemm_on() and PREPARE_MANY:    # the weight planes of all layers in three launches (instead of four per layer)
            grad = torch.is_grad_enabled()
            names = [f"conv{i}" for i in range(1, len(self.ENC) + 1)] + [name for name, _, _, _ in self.DEC]
            items = [(getattr(self, nm).weight, 2 if nm.startswith("conv") else 1,
                      grad and (nm != "conv1" or input.requires_grad)) for nm in names]
            planes = dict(zip(names, ops.conv2d_prepare_many(items)))
        skips = []
        for i in range(1, len(self.ENC) + 1):
            conv = getattr(self, f"conv{i}")
            x = _conv_block(x, None, conv.weight, conv.bias, 4, 2, ops.ACT_LEAKY if i > 1 else ops.ACT_NONE, False,
                            planes=planes.get(f"conv{i}"))
            bn = self.ENC_BN[i - 1]
            if bn is not None:
                m = getattr(self, bn)
                x = _BN2dFn.apply(x, m.weight, m.bias, m)
            skips.append(x)
        d0, d1 = skips.pop(), None                                                # innermost code: no skip of itself
        for name, _, _, bn in self.DEC:
            conv = getattr(self, name)
            d = _conv_block(d0, d1, conv.weight, conv.bias, 3, 1, ops.ACT_RELU, True, planes=planes.get(name))   # cat(d0, d1) never materialised
            if bn is not None:
                m = getattr(self, bn)
                d0, d1 = _BN2dFn.apply(d, m.weight, m.bias, m), skips.pop()
            else:
                d0, d1 = d, None
        return d0.permute(0, 3, 1, 2)                                             # (B,H,W,Cout) -> (B,Cout,H,W) view

    def _forward_stock(self, input):
        skips = []
        x = input
        for i in range(1, len(self.ENC) + 1):
            if i > 1:
                x = self.leaky_relu(x)
            x = getattr(self, f"conv{i}")(x)
            bn = self.ENC_BN[i - 1]
            if bn is not None:
                x = getattr(self, bn)(x)
            skips.append(x)
        d = skips.pop()
        for name, _, _, bn in self.DEC:
            d = getattr(self, name)(self.up(self.relu(d)))
            if bn is not None:
                d = torch.cat((getattr(self, bn)(d), skips.pop()), 1)
        return d


class Unet(_UNetBase):
    """8-down / 8-up regressor for 256x256 (or multiples) inputs."""
    ENC = (1, 2, 4, 8, 8, 8, 8, 8)
    ENC_BN = (None, "batch_norm2_0", "batch_norm4_0", "batch_norm8_0", "batch_norm8_1", "batch_norm8_2", "batch_norm8_3", None)
    DEC = (("dconv1", 8, 8, "batch_norm8_4"), ("dconv2", 16, 8, "batch_norm8_5"), ("dconv3", 16, 8, "batch_norm8_6"),
           ("dconv4", 16, 8, "batch_norm8_7"), ("dconv5", 16, 4, "batch_norm4_1"), ("dconv6", 8, 2, "batch_norm2_1"),
           ("dconv7", 4, 1, "batch_norm"), ("dconv8", 2, None, None))
    BN = (("batch_norm", 1), ("batch_norm2_0", 2), ("batch_norm2_1", 2), ("batch_norm4_0", 4), ("batch_norm4_1", 4),
          ("batch_norm8_0", 8), ("batch_norm8_1", 8), ("batch_norm8_2", 8), ("batch_norm8_3", 8), ("batch_norm8_4", 8),
          ("batch_norm8_5", 8), ("batch_norm8_6", 8), ("batch_norm8_7", 8))


class UNetMini(_UNetBase):
    """4-down / 4-up variant for un-resized 240x320 inputs."""
    ENC = (1, 2, 4, 8)
    ENC_BN = (None, "batch_norm2_0", "batch_norm4_0", None)
    DEC = (("dconv5", 8, 4, "batch_norm4_1"), ("dconv6", 8, 2, "batch_norm2_1"), ("dconv7", 4, 1, "batch_norm"),
           ("dconv8", 2, None, None))
    BN = (("batch_norm", 1), ("batch_norm2_0", 2), ("batch_norm2_1", 2), ("batch_norm4_0", 4), ("batch_norm4_1", 4))
