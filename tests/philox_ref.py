"""numpy replay of the device Philox-4x32-10 generators (fl_aux_kernels.hip)."""
import numpy as np

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c, k0, k1):
    c = [np.asarray(v, dtype=np.uint64) for v in c]
    k0 = np.uint64(k0)
    k1 = np.uint64(k1)
    for _ in range(10):
        p0 = np.uint64(M0) * c[0]
        p1 = np.uint64(M1) * c[2]
        n0 = ((p1 >> np.uint64(32)) ^ c[1] ^ k0) & MASK
        n1 = p1 & MASK
        n2 = ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & MASK
        n3 = p0 & MASK
        c = [n0, n1, n2, n3]
        k0 = (k0 + np.uint64(W0)) & MASK
        k1 = (k1 + np.uint64(W1)) & MASK
    return c


def u01(hi, lo):
    m = (hi << np.uint64(21)) | (lo >> np.uint64(11))
    return (m.astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def philox_uniform(seed, batch, n, lo, hi):
    pairs = (n + 1) // 2
    k, q = np.meshgrid(np.arange(batch, dtype=np.uint64), np.arange(pairs, dtype=np.uint64), indexing="ij")
    z = np.zeros_like(k)
    c = philox4x32_10([q, z, k, z], seed & 0xFFFFFFFF, seed >> 32)
    out = np.empty((batch, 2 * pairs))
    out[:, 0::2] = lo + (hi - lo) * u01(c[0], c[1])
    out[:, 1::2] = lo + (hi - lo) * u01(c[2], c[3])
    return out[:, :n].copy()


def philox_spectrum(seed, batch, n, klo, khi):
    k = np.arange(batch, dtype=np.uint64)
    z = np.zeros_like(k)
    c = philox4x32_10([z, z + np.uint64(1), k, z + np.uint64(1)], seed & 0xFFFFFFFF, seed >> 32)
    kappa = np.exp(np.log(klo) + u01(c[0], c[1]) * (np.log(khi) - np.log(klo)))
    i = np.arange(n, dtype=np.float64) / float(max(n - 1, 1))
    return 1.0 + (kappa[:, None] - 1.0) * i[None, :]
