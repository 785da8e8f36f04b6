"""Test infrastructure: the reference's full-batch iteration with every SUM carried in float64.

Used to attribute deviations: the oracle reproduces the reference bit for bit, INCLUDING the rounding noise of its
float32 sequential accumulations (numpy's row-by-row `sum(axis=0)`, the serial COO-order shape scatter), which reaches
1e-4 relative on rows with 10^4 nonzeros or tables with 10^4..10^6 rows.  The HIP path sums in trees / fp64 and is
compared with this variant where the reference's own noise would mask everything else."""
import numpy as np
import scipy.special as sp

from oracle import hpf_oracle as O


def exact_sums_reference(Y, iu, ii, nU, nI, k, its, seed=123):
    """The reference's iteration (PXI:227-259) with every SUM carried in float64 (phi normaliser, shape scatter, column
    and row sums) and float32 storage as in the reference: what the reference would give without its float32
    accumulation noise.  Pure numpy, small problems only."""
    st = O.State(nU, nI, O.Hyper(k, 0.3, 0.3, 1.0, 0.3, 0.3, 1.0), seed)
    hy = O.Hyper(k, 0.3, 0.3, 1.0, 0.3, 0.3, 1.0)
    iu = iu.astype(np.int64)
    ii = ii.astype(np.int64)
    f32, f64 = np.float32, np.float64
    for _ in range(its):
        e = (sp.psi(st.Gamma_shp[iu].astype(f64)) - np.log(st.Gamma_rte[iu].astype(f64))
             + sp.psi(st.Lambda_shp[ii].astype(f64)) - np.log(st.Lambda_rte[ii].astype(f64)))
        p = np.exp(e).astype(f32).astype(f64)
        phi = p * (Y.astype(f64) / p.sum(axis=1))[:, None]
        st.Gamma_rte = (f32(hy.k_shp) / st.k_rte + st.Beta.sum(axis=0, keepdims=True, dtype=f64).astype(f32)).astype(f32)
        G = np.full((nU, k), f64(hy.a))
        L = np.full((nI, k), f64(hy.c))
        np.add.at(G, iu, phi)
        np.add.at(L, ii, phi)
        st.Gamma_shp, st.Lambda_shp = G.astype(f32), L.astype(f32)
        st.Theta[:, :] = st.Gamma_shp / st.Gamma_rte
        st.Lambda_rte = (f32(hy.t_shp) / st.t_rte + st.Theta.sum(axis=0, keepdims=True, dtype=f64).astype(f32)).astype(f32)
        st.Beta[:, :] = st.Lambda_shp / st.Lambda_rte
        st.k_rte = (f32(hy.add_k_rte) + st.Theta.sum(axis=1, keepdims=True, dtype=f64).astype(f32)).astype(f32)
        st.t_rte = (f32(hy.add_t_rte) + st.Beta.sum(axis=1, keepdims=True, dtype=f64).astype(f32)).astype(f32)
    return st
