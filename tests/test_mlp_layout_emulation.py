"""CPU check of the MFMA index algebra in csrc/mlp.hip (forward and backward, f32 path).

The kernel body is replayed lane by lane with tests/emu/mfma_emulator.py for one 16-sample
wave tile and compared with a plain numpy MLP forward/backward.  This does not execute any
product code; it pins the layout reasoning the HIP kernel is written from."""
import numpy as np

from tests.emu.mfma_emulator import mfma_16x16x4_f32 as mfma

IN, HID, OUTP = 32, 64, 16
LDX, LDH = IN + 1, HID + 1


def _ref(x, W1, b1, W2, b2, W3, b3, dz3):
    z1 = x @ W1.T + b1
    a1 = np.maximum(z1, 0)
    z2 = a1 @ W2.T + b2
    a2 = np.maximum(z2, 0)
    h = a2 @ W3.T + b3
    da2 = dz3 @ W3
    dz2 = da2 * (a2 > 0)
    da1 = dz2 @ W2
    dz1 = da1 * (a1 > 0)
    dx = dz1 @ W1
    return h, dict(dW3=dz3.T @ a2, db3=dz3.sum(0), dW2=dz2.T @ a1, db2=dz2.sum(0), dW1=dz1.T @ x,
                   db1=dz1.sum(0), dx=dx)


def test_forward_and_backward_tile_layout():
    rng = np.random.RandomState(0)
    out_dim = 5
    x = rng.randn(16, IN)
    W1, b1 = rng.randn(HID, IN) * 0.3, rng.randn(HID) * 0.1
    W2, b2 = rng.randn(HID, HID) * 0.2, rng.randn(HID) * 0.1
    W3, b3 = rng.randn(out_dim, HID) * 0.2, rng.randn(out_dim) * 0.1
    dz3 = rng.randn(16, out_dim)
    h_ref, g = _ref(x, W1, b1, W2, b2, W3, b3, dz3)

    lanes = np.arange(64)
    j, q = lanes & 15, lanes >> 4
    # ---------------- forward, as k_mlp_forward_f32
    act = np.zeros(16 * LDH)
    acc = [np.repeat(b1[nt * 16 + j][:, None], 4, 1) for nt in range(4)]
    for kk in range(8):
        xa = x[j, 4 * kk + q]  # load_feat(level 2kk + (q>>1), f = q&1) == column 4kk+q
        for nt in range(4):
            acc[nt] = mfma(xa, W1[nt * 16 + j, 4 * kk + q], acc[nt])
    a1c = [np.maximum(a, 0) for a in acc]
    for nt in range(4):
        for r in range(4):
            act[(q * 4 + r) * LDH + nt * 16 + j] = a1c[nt][:, r]
    P = act.copy()
    acc = [np.repeat(b2[nt * 16 + j][:, None], 4, 1) for nt in range(4)]
    for kk in range(16):
        xa = act[j * LDH + 4 * kk + q]
        for nt in range(4):
            acc[nt] = mfma(xa, W2[nt * 16 + j, 4 * kk + q], acc[nt])
    a2c = [np.maximum(a, 0) for a in acc]
    for nt in range(4):
        for r in range(4):
            act[(q * 4 + r) * LDH + nt * 16 + j] = a2c[nt][:, r]
    Q = act.copy()
    W3p = np.zeros((OUTP, HID)); W3p[:out_dim] = W3
    b3p = np.zeros(OUTP); b3p[:out_dim] = b3
    o = np.repeat(b3p[j][:, None], 4, 1)
    for kk in range(16):
        o = mfma(act[j * LDH + 4 * kk + q], W3p[j, 4 * kk + q], o)
    h = np.zeros((16, OUTP))
    for l in range(64):
        for r in range(4):
            h[q[l] * 4 + r, j[l]] = o[l, r]
    np.testing.assert_allclose(h[:, :out_dim], h_ref, rtol=1e-9, atol=1e-9)

    # ---------------- backward, as k_mlp_backward_f32
    X = np.zeros(16 * LDX)
    for kk in range(8):
        X[j * LDX + 4 * kk + q] = x[j, 4 * kk + q]
    sW1 = np.zeros(HID * LDX); sW2 = np.zeros(HID * LDH); sW3 = np.zeros(OUTP * LDH)
    for o_ in range(HID):
        sW1[o_ * LDX:o_ * LDX + IN] = W1[o_]
        sW2[o_ * LDH:o_ * LDH + HID] = W2[o_]
    for o_ in range(out_dim):
        sW3[o_ * LDH:o_ * LDH + HID] = W3[o_]
    z0 = lambda: np.zeros((64, 4))
    # dz3 in A layout + db3
    dz3a = []
    gb3 = []
    for kk in range(2):
        k = 4 * kk + q
        v = np.where(k < out_dim, dz3[j, np.minimum(k, out_dim - 1)], 0.0)
        dz3a.append(v)
        gb3.append(v.copy())
    # dW3
    gW3 = [z0() for _ in range(4)]
    for kk in range(4):
        s = 4 * kk + q
        v = np.where(j < out_dim, dz3[s, np.minimum(j, out_dim - 1)], 0.0)
        for nt in range(4):
            gW3[nt] = mfma(v, Q[(4 * kk + q) * LDH + nt * 16 + j], gW3[nt])
    dW3 = np.zeros((OUTP, HID))
    for nt in range(4):
        for l in range(64):
            for r in range(4):
                dW3[q[l] * 4 + r, nt * 16 + j[l]] = gW3[nt][l, r]
    np.testing.assert_allclose(dW3[:out_dim], g["dW3"], rtol=1e-9, atol=1e-9)
    assert np.abs(dW3[out_dim:]).max() == 0
    # dA2 / dZ2
    z = [z0() for _ in range(4)]
    for kk in range(2):
        for nt in range(4):
            z[nt] = mfma(dz3a[kk], sW3[(4 * kk + q) * LDH + nt * 16 + j], z[nt])
    gb2 = [np.zeros(64) for _ in range(4)]
    for nt in range(4):
        for r in range(4):
            d = np.where(a2c[nt][:, r] > 0, z[nt][:, r], 0.0)
            Q[(q * 4 + r) * LDH + nt * 16 + j] = d
            gb2[nt] += d
    # dW2
    gW2 = [[z0() for _ in range(4)] for _ in range(4)]
    for kk in range(4):
        av = [Q[(4 * kk + q) * LDH + t * 16 + j] for t in range(4)]
        bv = [P[(4 * kk + q) * LDH + t * 16 + j] for t in range(4)]
        for mt in range(4):
            for nt in range(4):
                gW2[mt][nt] = mfma(av[mt], bv[nt], gW2[mt][nt])
    dW2 = np.zeros((HID, HID))
    for mt in range(4):
        for nt in range(4):
            for l in range(64):
                for r in range(4):
                    dW2[mt * 16 + q[l] * 4 + r, nt * 16 + j[l]] = gW2[mt][nt][l, r]
    np.testing.assert_allclose(dW2, g["dW2"], rtol=1e-9, atol=1e-9)
    # dA1 / dZ1
    z = [z0() for _ in range(4)]
    for kk in range(16):
        xa = Q[j * LDH + 4 * kk + q]
        for nt in range(4):
            z[nt] = mfma(xa, sW2[(4 * kk + q) * LDH + nt * 16 + j], z[nt])
    gb1 = [np.zeros(64) for _ in range(4)]
    for nt in range(4):
        for r in range(4):
            d = np.where(a1c[nt][:, r] > 0, z[nt][:, r], 0.0)
            P[(q * 4 + r) * LDH + nt * 16 + j] = d
            gb1[nt] += d
    # dW1
    gW1 = [[z0(), z0()] for _ in range(4)]
    for kk in range(4):
        av = [P[(4 * kk + q) * LDH + t * 16 + j] for t in range(4)]
        bv = [X[(4 * kk + q) * LDX + t * 16 + j] for t in range(2)]
        for mt in range(4):
            for nt in range(2):
                gW1[mt][nt] = mfma(av[mt], bv[nt], gW1[mt][nt])
    dW1 = np.zeros((HID, IN))
    for mt in range(4):
        for nt in range(2):
            for l in range(64):
                for r in range(4):
                    dW1[mt * 16 + q[l] * 4 + r, nt * 16 + j[l]] = gW1[mt][nt][l, r]
    np.testing.assert_allclose(dW1, g["dW1"], rtol=1e-9, atol=1e-9)
    # dX
    dx = [z0(), z0()]
    for kk in range(16):
        xa = P[j * LDH + 4 * kk + q]
        dx[0] = mfma(xa, sW1[(4 * kk + q) * LDX + j], dx[0])
        dx[1] = mfma(xa, sW1[(4 * kk + q) * LDX + 16 + j], dx[1])
    dX = np.zeros((16, IN))
    for nt in range(2):
        for l in range(64):
            for r in range(4):
                dX[q[l] * 4 + r, nt * 16 + j[l]] = dx[nt][l, r]
    np.testing.assert_allclose(dX, g["dx"], rtol=1e-9, atol=1e-9)
    # biases: xor-shuffle reductions
    for nt in range(4):
        v1 = gb1[nt].reshape(4, 16).sum(0)
        v2 = gb2[nt].reshape(4, 16).sum(0)
        np.testing.assert_allclose(v1, g["db1"][nt * 16:(nt + 1) * 16], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(v2, g["db2"][nt * 16:(nt + 1) * 16], rtol=1e-9, atol=1e-9)
    db3 = np.zeros(8)
    for kk in range(2):
        v = gb3[kk].reshape(4, 16).sum(1)  # sum over j inside each q group
        for qq in range(4):
            db3[4 * kk + qq] = v[qq]
    np.testing.assert_allclose(db3[:out_dim], g["db3"], rtol=1e-9, atol=1e-9)
