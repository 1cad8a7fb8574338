"""Numpy emulation of v_mfma_f32_16x16x4_f32 and of the jet kernels' fragment layouts (host-logic tests only)."""
import numpy as np

LANE = np.arange(64)
G, J = LANE >> 4, LANE & 15


def mfma4(a, b, c):
    """a[64], b[64], c[64,4] -> d[64,4]: D[i][j] = sum_k A[i][k] B[k][j] + C, A[i][k]=a[16k+i], B[k][j]=b[16k+j],
    d[16g+j][r] = D[4g+r][j]."""
    A = a.reshape(4, 16).T          # [i][k]
    B = b.reshape(4, 16)            # [k][j]
    D = A @ B                       # [i][j]
    d = np.empty((64, 4), dtype=np.float64)
    for r in range(4):
        d[:, r] = D[4 * G + r, J]
    return d + c


def to_frag(mat):
    """mat [16 rows, 16*T features] -> fragment blocks [T][64][4] (lane 16g+j: features 4g..4g+3 of row j)."""
    T = mat.shape[1] // 16
    out = np.empty((T, 64, 4))
    for t in range(T):
        for r in range(4):
            out[t, :, r] = mat[J, 16 * t + 4 * G + r]
    return out


def from_frag(frag):
    T = frag.shape[0]
    mat = np.empty((16, 16 * T))
    for t in range(T):
        for r in range(4):
            mat[J, 16 * t + 4 * G + r] = frag[t, :, r]
    return mat


def gemm_frag(wpack, bfrag, KT, MT):
    """Emulates k_layer's main loop: out[mt] = sum_kt sum_r mfma4(wpack[kt][mt][:, r], bfrag[kt][:, r])."""
    out = np.zeros((MT, 64, 4))
    for kt in range(KT):
        for mt in range(MT):
            for r in range(4):
                out[mt] = mfma4(wpack[kt, mt, :, r], bfrag[kt, :, r], out[mt])
    return out


def transpose_block(v):
    """k_wgrad's LDS transpose: D-image [64,4] -> operand image out[64, 4 steps] (lane 16k+i, step s: row 4s+k, feat i)."""
    patch = np.empty((16, 16))
    for r in range(4):
        patch[J, 4 * G + r] = v[:, r]
    out = np.empty((64, 4))
    for s in range(4):
        out[:, s] = patch[4 * s + G, J]
    return out
