"""The opt-in `blocked` layout of the hashed levels (include/lnerf_hip.h LNERF_GRID_BLOCKED), oracle side: index
properties that the kernels rely on, checked on the CPU."""
import torch

from oracle import nerf_oracle as O


def test_blocked_indices_keep_blocks_together_and_in_range():
    hs = 1 << 19
    res = 1024                                   # (res + 1)^3 > 2^19: a hashed level
    g = torch.Generator().manual_seed(0)
    pos = torch.randint(0, res + 1, (20000, 3), generator=g)
    idx = O.grid_corner_indices(pos, res, hs, blocked=True)
    assert int(idx.min()) >= 0 and int(idx.max()) < hs
    # the 16 vertices of one 4 x 2 x 2 block occupy 16 consecutive rows starting at a multiple of 16
    base = (pos >> torch.tensor([2, 1, 1])) << torch.tensor([2, 1, 1])
    rows = []
    for dz in range(2):
        for dy in range(2):
            for dx in range(4):
                rows.append(O.grid_corner_indices(base + torch.tensor([dx, dy, dz]), res, hs, blocked=True))
    rows = torch.stack(rows, 1)
    assert torch.equal(rows - rows[:, :1], torch.arange(16)[None].expand_as(rows))
    assert int((rows[:, 0] % 16).abs().max()) == 0
    # dense levels are untouched by the flag
    small = torch.randint(0, 17, (1000, 3), generator=g)
    assert torch.equal(O.grid_corner_indices(small, 16, hs, blocked=True), O.grid_corner_indices(small, 16, hs))
    # and the blocked hash differs from the vertex hash (it is a different table layout)
    assert not torch.equal(idx, O.grid_corner_indices(pos, res, hs))


def test_blocked_encode_is_differentiable_and_lines_per_sample_drop():
    lv = O.make_grid_levels(blocked=True)
    lv0 = O.make_grid_levels()
    assert lv.offsets == lv0.offsets and lv.blocked and not lv0.blocked
    g = torch.Generator().manual_seed(1)
    x = torch.rand(256, 3, generator=g)
    table = (torch.randn(lv.n_rows, 2, generator=g) * 0.1).requires_grad_()
    feat = O.grid_encode(x, table, lv)
    feat.sum().backward()
    assert feat.shape == (256, 32) and float(table.grad.abs().sum()) > 0
    # 64-byte lines (16 bf16 rows) touched per sample on the finest level: blocked < vertex hash
    def lines(levels):
        l = levels.num_levels - 1
        pos = x * levels.scales[l] + 0.5
        pg = torch.floor(pos).to(torch.int64)
        hs = levels.offsets[l + 1] - levels.offsets[l]
        n = 0
        for i in range(x.shape[0]):
            rows = set()
            for c in range(8):
                corner = pg[i] + torch.tensor([c & 1, (c >> 1) & 1, (c >> 2) & 1])
                rows.add(int(O.grid_corner_indices(corner[None], levels.resolutions[l], hs, levels.blocked)[0]) // 16)
            n += len(rows)
        return n / x.shape[0]
    a, b = lines(lv0), lines(lv)
    assert 3.9 < a < 4.6 and 2.4 < b < 3.2, (a, b)


def test_tiled_indices_wrap_the_dense_index():
    """gridtype = "tiled" (the upstream encoder's other layout): a level too large for its table wraps its dense index
    -- x-neighbours stay neighbours (mod table size), whole slabs alias -- and dense levels are unchanged."""
    res, hs = 100, 2 ** 12
    g = torch.Generator().manual_seed(0)
    pos = torch.randint(0, res, (2000, 3), generator=g)
    idx = O.grid_corner_indices(pos, res, hs, tiled=True)
    assert int(idx.min()) >= 0 and int(idx.max()) < hs
    st = res + 1
    assert torch.equal(idx, (pos[:, 0] + pos[:, 1] * st + pos[:, 2] * st * st) % hs)
    nxt = O.grid_corner_indices(pos + torch.tensor([1, 0, 0]), res, hs, tiled=True)
    assert torch.equal(nxt, (idx + 1) % hs)
    small = torch.randint(0, 16, (100, 3), generator=g)
    assert torch.equal(O.grid_corner_indices(small, 16, hs * 4, tiled=True), O.grid_corner_indices(small, 16, hs * 4))
    assert not torch.equal(idx, O.grid_corner_indices(pos, res, hs))
    # 32-bit wrap of the dense index on the finest default level (2049^3 > 2^32), then the table size
    big = torch.tensor([[2047, 2047, 2047]])
    want = ((2047 + 2047 * 2049 + 2047 * 2049 * 2049) & 0xFFFFFFFF) % (2 ** 19)
    assert int(O.grid_corner_indices(big, 2048, 2 ** 19, tiled=True)[0]) == want
    lv = O.make_grid_levels(tiled=True)
    assert lv.tiled and lv.offsets == O.make_grid_levels().offsets
