"""CPU replay of the index algebra of the bf16 MFMA MLP (csrc/mlp_bf16.hip): the "sample on the
lane" formulation Z^T = W X^T in which every layer's accumulator tile is the next MFMA's B operand
(k order permuted by phi), and the weight gradients that go through [sample][feature] LDS images written
with the packed fragment halves and read back with the transposing LDS read (ds_read_b64_tr_b16).
Pure numpy in float64 -- this pins layouts, not rounding."""
import numpy as np

from tests.emu.mfma_emulator import ds_read_tr16_b64
from tests.emu.mfma_emulator import mfma_16x16x32 as mfma

IN, HID = 32, 64
L = np.arange(64)
Q, C = L >> 4, L & 15


def phi(s, q, jj):
    return 32 * s + 16 * (jj >> 2) + 4 * q + (jj & 3)


def frag(fn):
    """[64, 8] fragment from fn(lane q, lane c, jj)."""
    out = np.zeros((64, 8))
    for l in range(64):
        for jj in range(8):
            out[l, jj] = fn(Q[l], C[l], jj)
    return out


def cfrag(fn):
    out = np.zeros((64, 4))
    for l in range(64):
        for r in range(4):
            out[l, r] = fn(Q[l], C[l], r)
    return out


def pack(tiles, s, t):
    """B fragment of k-step s from the C-layout tiles (2s, 2s+1) of column tile t."""
    out = np.zeros((64, 8))
    for jj in range(8):
        out[:, jj] = tiles[2 * s + (jj >> 2)][t][:, jj & 3]
    return out


def test_bf16_formulation_forward_backward():
    rng = np.random.RandomState(0)
    out_dim, S = 5, 32  # 32 samples = two column tiles
    X = rng.randn(S, IN)
    W1, b1 = rng.randn(HID, IN) * 0.3, rng.randn(HID) * 0.1
    W2, b2 = rng.randn(HID, HID) * 0.2, rng.randn(HID) * 0.1
    W3, b3 = rng.randn(out_dim, HID) * 0.2, rng.randn(out_dim) * 0.1
    dZ3 = rng.randn(S, out_dim)
    # reference
    Z1 = X @ W1.T + b1; A1 = np.maximum(Z1, 0)
    Z2 = A1 @ W2.T + b2; A2 = np.maximum(Z2, 0)
    Hh = A2 @ W3.T + b3
    dA2 = dZ3 @ W3; dZ2 = dA2 * (A2 > 0)
    dA1 = dZ2 @ W2; dZ1 = dA1 * (A1 > 0)
    dX = dZ1 @ W1
    ref = dict(dW3=dZ3.T @ A2, dW2=dZ2.T @ A1, dW1=dZ1.T @ X, db3=dZ3.sum(0), db2=dZ2.sum(0), db1=dZ1.sum(0))

    # ---------------- forward
    w1a = [frag(lambda q, c, jj, mt=mt: W1[16 * mt + c, 8 * q + jj]) for mt in range(4)]
    xB = [frag(lambda q, c, jj, t=t: X[16 * t + c, 8 * q + jj]) for t in range(2)]
    bias1 = [cfrag(lambda q, c, r, mt=mt: b1[16 * mt + 4 * q + r]) for mt in range(4)]
    acc1 = [[mfma(w1a[mt], xB[t], bias1[mt]) for t in range(2)] for mt in range(4)]
    for mt in range(4):
        for t in range(2):
            for l in range(64):
                for r in range(4):
                    assert abs(acc1[mt][t][l, r] - Z1[16 * t + C[l], 16 * mt + 4 * Q[l] + r]) < 1e-9
    h1 = [[np.maximum(acc1[mt][t], 0) for t in range(2)] for mt in range(4)]
    w2a = [[frag(lambda q, c, jj, mt=mt, s=s: W2[16 * mt + c, phi(s, q, jj)]) for s in range(2)] for mt in range(4)]
    bias2 = [cfrag(lambda q, c, r, mt=mt: b2[16 * mt + 4 * q + r]) for mt in range(4)]
    acc2 = [[None, None] for _ in range(4)]
    for mt in range(4):
        for t in range(2):
            a = bias2[mt]
            for s in range(2):
                a = mfma(w2a[mt][s], pack(h1, s, t), a)
            acc2[mt][t] = a
    h2 = [[np.maximum(acc2[mt][t], 0) for t in range(2)] for mt in range(4)]
    w3a = [frag(lambda q, c, jj, s=s: W3[c, phi(s, q, jj)] if c < out_dim else 0.0) for s in range(2)]
    bias3 = cfrag(lambda q, c, r: b3[4 * q + r] if 4 * q + r < out_dim else 0.0)
    for t in range(2):
        o = bias3
        for s in range(2):
            o = mfma(w3a[s], pack(h2, s, t), o)
        for l in range(64):
            for r in range(4):
                n = 4 * Q[l] + r
                if n < out_dim:
                    assert abs(o[l, r] - Hh[16 * t + C[l], n]) < 1e-9

    # ---------------- backward data chain
    z4 = np.zeros((64, 4))
    dz3B = [frag(lambda q, c, jj, t=t: dZ3[16 * t + c, 4 * q + jj] if (jj < 4 and 4 * q + jj < out_dim) else 0.0)
            for t in range(2)]
    w3ta = [frag(lambda q, c, jj, mt=mt: W3[4 * q + jj, 16 * mt + c] if (jj < 4 and 4 * q + jj < out_dim) else 0.0)
            for mt in range(4)]
    dz2 = [[mfma(w3ta[mt], dz3B[t], z4) * (acc2[mt][t] > 0) for t in range(2)] for mt in range(4)]
    w2ta = [[frag(lambda q, c, jj, mt=mt, s=s: W2[phi(s, q, jj), 16 * mt + c]) for s in range(2)] for mt in range(4)]
    dz1 = [[None, None] for _ in range(4)]
    for mt in range(4):
        for t in range(2):
            a = z4
            for s in range(2):
                a = mfma(w2ta[mt][s], pack(dz2, s, t), a)
            dz1[mt][t] = a * (acc1[mt][t] > 0)
    w1ta = [[frag(lambda q, c, jj, mt=mt, s=s: W1[phi(s, q, jj), 16 * mt + c]) for s in range(2)] for mt in range(2)]
    for mt in range(2):
        for t in range(2):
            a = z4
            for s in range(2):
                a = mfma(w1ta[mt][s], pack(dz1, s, t), a)
            for l in range(64):
                for r in range(4):
                    assert abs(a[l, r] - dX[16 * t + C[l], 16 * mt + 4 * Q[l] + r]) < 1e-9

    # ---------------- weight gradients through [sample][feature] LDS images (one wave, 32 samples = one k-step)
    RS = 68

    def stage_pair(img, t, fa_of, fb_of, fragB):
        """kernel stage_pair: lane (q, c) stores elements 0..3 at features fa..fa+3 and 4..7 at fb..fb+3 of its
        sample row 16t + c."""
        for l in range(64):
            row = 16 * t + C[l]
            img[row, fa_of(Q[l]):fa_of(Q[l]) + 4] = fragB[l, 0:4]
            if fb_of is not None:
                img[row, fb_of(Q[l]):fb_of(Q[l]) + 4] = fragB[l, 4:8]

    def image(tiles):
        img = np.full((S, RS), np.nan)
        for t in range(2):
            for s in range(2):
                stage_pair(img, t, lambda q, s=s: 32 * s + 4 * q, lambda q, s=s: 32 * s + 16 + 4 * q, pack(tiles, s, t))
        return img

    ldsH1, ldsH2 = image(h1), image(h2)
    ldsD1, ldsD2 = image(dz1), image(dz2)
    ldsX = np.full((S, RS), np.nan)
    for t in range(2):  # lane (q,c) holds input features 8q..8q+7 of sample 16t+c
        stage_pair(ldsX, t, lambda q: 8 * q, lambda q: 8 * q + 4, xB[t])
    ldsD3 = np.full((S, RS), np.nan)
    for t in range(2):  # dZ3: slot (q, jj < 4) <-> output 4q + jj: only the low half is stored
        stage_pair(ldsD3, t, lambda q: 4 * q, None, dz3B[t])

    def ld_tr(img, k, f0):
        """kernel ld_tr: feature f0 + (l & 15) on the lane, samples 32k + 8(l >> 4) + 0..7 in the 8 elements."""
        rows = 32 * k + 8 * Q + (C >> 2)
        cols = f0 + 4 * (C & 3)
        lo = ds_read_tr16_b64(img, rows, cols)
        hi = ds_read_tr16_b64(img, rows + 4, cols)
        return np.concatenate([lo, hi], axis=1)

    def dw(ldsD, ldsA, mt, nt):
        a = ld_tr(ldsD, 0, 16 * mt)                                   # A[i = feature row][k = sample 8q+jj]
        b = ld_tr(ldsA, 0, 16 * nt)                                   # B[k = sample][col = feature]
        assert not np.isnan(a).any() and not np.isnan(b).any()        # only staged elements are ever read
        # and the fragments are what the old [feature][sample] formulation asked for
        for l in range(64):
            for jj in range(8):
                assert a[l, jj] == ldsD[8 * Q[l] + jj, 16 * mt + C[l]] and b[l, jj] == ldsA[8 * Q[l] + jj, 16 * nt + C[l]]
        return mfma(a, b, z4)                                         # K = 32 samples in one step

    def check(ldsD, ldsA, n_mt, n_nt, want):
        for mt in range(n_mt):
            for nt in range(n_nt):
                acc = dw(ldsD, ldsA, mt, nt)
                for l in range(64):
                    for r in range(4):
                        o, i = 16 * mt + 4 * Q[l] + r, 16 * nt + C[l]
                        if o < want.shape[0]:
                            assert abs(acc[l, r] - want[o, i]) < 1e-9

    check(ldsD2, ldsH1, 4, 4, ref["dW2"])
    check(ldsD1, ldsX, 4, 2, ref["dW1"])
    check(ldsD3, ldsH2, 1, 4, ref["dW3"])
    # bias gradients with a B tile of ones: every column of the result is the row sum
    ones = np.ones((64, 8))
    for mt in range(4):
        a = ld_tr(ldsD2, 0, 16 * mt)
        acc = mfma(a, ones, z4)
        for l in range(64):
            for r in range(4):
                assert abs(acc[l, r] - ref["db2"][16 * mt + 4 * Q[l] + r]) < 1e-9
