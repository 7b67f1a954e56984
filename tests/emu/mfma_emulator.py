"""Lane-level emulation of the gfx950 MFMA shapes used by csrc/mlp.hip.

Used only by CPU tests: it replays the kernels' index algebra (which lane holds which matrix
element, what goes where in LDS) with numpy so that layout mistakes are caught without a GPU.
Lane maps (cdna_hip_programming.md §3):
  v_mfma_f32_16x16x4_f32 :  A[i = l&15][k = l>>4]   B[k = l>>4][j = l&15]
  C/D (all 16x16 shapes) :  D[i = (l>>4)*4 + reg][j = l&15]
"""
import numpy as np


def mfma_16x16x4_f32(a, b, c):
    """a, b: [64] per-lane scalars; c: [64,4] per-lane accumulators -> new c."""
    A = np.zeros((16, 4), dtype=np.float64)
    B = np.zeros((4, 16), dtype=np.float64)
    for l in range(64):
        A[l & 15, l >> 4] = a[l]
        B[l >> 4, l & 15] = b[l]
    D = A @ B
    out = c.astype(np.float64).copy()
    for l in range(64):
        for r in range(4):
            out[l, r] += D[(l >> 4) * 4 + r, l & 15]
    return out


def mfma_16x16x32(a, b, c):
    """v_mfma_f32_16x16x32_{bf16,f16} lane maps: a, b: [64, 8] per-lane fragments
    (A[i = l&15][k = 8*(l>>4) + jj], B[k = 8*(l>>4) + jj][j = l&15]); c: [64, 4]."""
    A = np.zeros((16, 32), dtype=np.float64)
    B = np.zeros((32, 16), dtype=np.float64)
    for l in range(64):
        for jj in range(8):
            A[l & 15, 8 * (l >> 4) + jj] = a[l, jj]
            B[8 * (l >> 4) + jj, l & 15] = b[l, jj]
    D = A @ B
    out = c.astype(np.float64).copy()
    for l in range(64):
        for r in range(4):
            out[l, r] += D[(l >> 4) * 4 + r, l & 15]
    return out


def ds_read_tr16_b64(lds, rows, cols):
    """gfx950 ds_read_b64_tr_b16 (cdna_hip_programming.md T10).  lds: 2-D array [row][16-bit element];
    rows/cols: [64] per-lane address (row, first of 4 consecutive columns).  Per group of 16 lanes, lane 4a+b of the
    group supplies the address of row a, columns 4b..4b+3 of a 4 x 16 block; lane i of the group receives column i
    of the 4 rows, row a in its element a.  Returns [64, 4]."""
    out = np.zeros((64, 4))
    for g in range(4):
        block = np.zeros((4, 16))
        for a in range(4):
            for b in range(4):
                src = 16 * g + 4 * a + b
                block[a, 4 * b:4 * b + 4] = lds[rows[src], cols[src]:cols[src] + 4]
        for i in range(16):
            out[16 * g + i, :] = block[:, i]
    return out
