"""Stochastic (mini-batch) paths: SVI epochs of fit_hpf (cython_loops.pxi:262-377), the Cython
partial_fit (cython_loops.pxi:423-473) and the single-user fold-in (cython_loops.pxi:476-520).

These are the "next" rows of SURVEY.md section 8f-1/8f-4; they reuse the sweep kernel with a
row-list of segments and the atomic scatter variant.  Not implemented yet in this round: the
functions raise instead of silently computing on the CPU.
"""


def _todo(what):
    raise NotImplementedError(
        "hpfrec_amd: %s is not implemented on the HIP path yet (SURVEY.md section 8f); "
        "full-batch fit (users_per_batch=None, items_per_batch=None) is." % what)


def fit_hpf_svi(*args, **kwargs):
    _todo("stochastic variational inference (users_per_batch / items_per_batch)")


def partial_fit_step(*args, **kwargs):
    _todo("partial_fit")


def calc_user_factors(*args, **kwargs):
    _todo("predict_factors / add_user fold-in")
