"""H11 background net (frequency encoding degree 6 -> 64 -> C), HIP-backed autograd Function."""
import torch

from ..raymarching import backend as _b
from ..raymarching.raymarching import _chk, _p, _stream


class _BgNet(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dirs, w1, b1, w2, b2):
        dirs = dirs.contiguous()
        N, C = dirs.shape[0], w2.shape[0]
        out = torch.empty(N, C, device=dirs.device, dtype=torch.float32)
        _b.call("lnerf_bg_forward", _chk(dirs, "dirs"), N, _chk(w1, "w1"), _chk(b1, "b1"), _chk(w2, "w2"),
                _chk(b2, "b2"), C, _p(out), _stream())
        ctx.save_for_backward(dirs, w1, b1, w2, b2)
        return out

    @staticmethod
    def backward(ctx, dout):
        dirs, w1, b1, w2, b2 = ctx.saved_tensors
        N, C = dirs.shape[0], w2.shape[0]
        grads = [torch.zeros_like(t) for t in (w1, b1, w2, b2)]
        _b.call("lnerf_bg_backward", _p(dirs), N, _p(w1), _p(b1), _p(w2), _p(b2), C,
                _chk(dout.contiguous(), "dout"), *[_p(g) for g in grads], _stream())
        return (None, *grads)


def background_net(dirs, w1, b1, w2, b2):
    return _BgNet.apply(dirs.reshape(-1, 3), w1, b1, w2, b2)
