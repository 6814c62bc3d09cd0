"""CPU validation of passl_amd/hip/plan.py: the descriptors + pack jobs, executed by the
emulator (tests/emu.py), must reproduce torch's conv2d forward / input-grad / weight-grad."""
import pytest
import torch
import torch.nn.functional as F

import emu
from passl_amd.hip import plan as P



@pytest.fixture(autouse=True)
def _fp64_default():
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    yield
    torch.set_default_dtype(old)

GEOMS = [
    P.ConvGeom(cin=8, cout=16, k=3, stride=1, pad=1),
    P.ConvGeom(cin=8, cout=16, k=3, stride=2, pad=1),
    P.ConvGeom(cin=16, cout=8, k=1, stride=1, pad=0),
    P.ConvGeom(cin=16, cout=8, k=1, stride=2, pad=0),
    P.ConvGeom(cin=8, cout=8, k=7, stride=2, pad=3),
    P.ConvGeom(cin=8, cout=8, k=5, stride=3, pad=2),
]
SIZES = [(2, 8, 8), (1, 7, 9), (2, 5, 6)]


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize('g', GEOMS)
@pytest.mark.parametrize('nhw', SIZES)
def test_fwd_dgrad_wgrad_plans(g, nhw):
    N, H, W = nhw
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(N, g.cin, H, W, generator=gen, requires_grad=True)
    w = torch.randn(g.cout, g.cin, g.k, g.k, generator=gen, requires_grad=True)
    y = F.conv2d(x, w, None, g.stride, g.pad)
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy)
    w_krsc = w.detach().permute(0, 2, 3, 1).contiguous()
    x_flat = _nhwc(x.detach()).reshape(-1)
    dy_nhwc = _nhwc(dy)
    # forward
    d = P.fwd_desc(g, N, H, W)
    yf = torch.zeros(N * d.OP * d.OQ * g.cout)
    emu.emu_conv(d, x_flat, emu.emu_pack(d.pack, w_krsc), yf)
    assert (d.OP, d.OQ) == tuple(y.shape[2:])
    torch.testing.assert_close(yf.view(N, d.OP, d.OQ, g.cout), _nhwc(y.detach()))
    # data gradient
    descs, skipped = P.dgrad_plan(g, N, H, W)
    dxf = torch.full((N * H * W * g.cin,), float('nan'))
    if skipped:
        dxf.zero_()
    for dd in descs:
        emu.emu_conv(dd, dy_nhwc.reshape(-1), emu.emu_pack(dd.pack, w_krsc), dxf)
    assert not torch.isnan(dxf).any(), 'dgrad classes do not cover dx'
    torch.testing.assert_close(dxf.view(N, H, W, g.cin), _nhwc(x.grad))
    # weight gradient
    wd = P.wgrad_desc(g, N, H, W)
    dw = emu.emu_wgrad(wd, x_flat, dy_nhwc.reshape(-1, g.cout))
    torch.testing.assert_close(dw.view(g.cout, g.k, g.k, g.cin), w.grad.permute(0, 2, 3, 1))


@pytest.mark.parametrize('hw', [(224, 224), (64, 64), (33, 47)])
def test_stem_plan(hw):
    H, W = hw
    N, cout = 2, 8
    gen = torch.Generator().manual_seed(2)
    x = torch.randn(N, 3, H, W, generator=gen, requires_grad=True)
    w = torch.randn(cout, 3, 7, 7, generator=gen, requires_grad=True)
    y = F.conv2d(x, w, None, 2, 3)
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy)
    Hp, Wp = P.stem_padded_hw(H, W)
    xp = torch.zeros(N, Hp, Wp, P.STEM_CP)
    xp[:, 3:3 + H, 3:3 + W, :3] = _nhwc(x.detach())
    d = P.stem_desc(cout, N, H, W)
    w_krsc = w.detach().permute(0, 2, 3, 1).contiguous()
    wp = emu.emu_pack(d.pack, w_krsc)                  # [cout,7,8,4]
    assert wp.shape == (cout, 7, 8, 4)
    yf = torch.zeros(N * d.OP * d.OQ * cout)
    emu.emu_conv(d, xp.reshape(-1), wp, yf)
    torch.testing.assert_close(yf.view(N, d.OP, d.OQ, cout), _nhwc(y.detach()))
    dwp = emu.emu_wgrad(d, xp.reshape(-1), _nhwc(dy).reshape(-1, cout)).view(cout, 7, 8, 4)
    torch.testing.assert_close(dwp[:, :, :7, :3], w.grad.permute(0, 2, 3, 1))


def test_wgrad_splits_bounds():
    assert P.wgrad_splits(802816, 64, 64, 64) >= 256
    assert P.wgrad_splits(64, 2048, 4608, 64) == 1


def test_wgrad_grid_targets_one_workgroup_per_cu_for_convolutions_two_for_linears():
    """Round 6 (profiles/r06_wgrad_grid_ab.txt): inside the step a convolutional layer's weight-gradient
    launch gets ~256 workgroups, a Linear's ~512; the slice count follows."""
    import torch
    from passl_amd.hip import ops
    assert P.wgrad_target_blocks(True) == 256 and P.wgrad_target_blocks(False) == 512
    g1 = P.ConvGeom(cin=64, cout=256, k=1, stride=1, pad=0)
    conv = P.wgrad_desc(g1, 256, 56, 56)                    # 64 -> 256 1x1 at 56 x 56: 2 tiles of dW
    lin = P.wgrad_desc(g1, 256 * 56 * 56, 1, 1)             # the same matrix product as a Linear over rows
    assert ops.wgrad_slices(conv, torch.bfloat16) == 128 and ops.wgrad_slices(lin, torch.bfloat16) == 256
    g3 = P.ConvGeom(cin=64, cout=64, k=3, stride=1, pad=1)
    assert ops.wgrad_slices(P.wgrad_desc(g3, 256, 56, 56), torch.bfloat16) == 256      # spatially tiled 3x3 kernel
    assert P.wgrad_halo_splits(256, 14, 14, 256, 256) == 16
