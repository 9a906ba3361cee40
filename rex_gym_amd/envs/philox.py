"""Philox4x32-10 in numpy (host-side mirror of the generator the kernels draw with, csrc/rexsim.hip `philox4x32`): lets
the host tell which task an env of a mixed batch runs, or which terrain / randomisation draw it got, without a launch."""
import numpy as np


def philox4x32(ctr, k0, k1):
    """ctr: uint32 [4, N]; k0, k1: uint32 scalars.  Returns the uint32 [4, N] output block."""
    c = [np.asarray(x, dtype=np.uint64) for x in ctr]
    k0 = np.uint64(k0); k1 = np.uint64(k1)
    m0, m1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = m0 * c[0], m1 * c[2]
        n0 = ((p1 >> np.uint64(32)) ^ c[1] ^ k0) & mask
        n1 = p1 & mask
        n2 = ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & mask
        n3 = p0 & mask
        c = [n0, n1, n2, n3]
        k0 = (k0 + np.uint64(0x9E3779B9)) & mask
        k1 = (k1 + np.uint64(0xBB67AE85)) & mask
    return np.stack(c).astype(np.uint32)
