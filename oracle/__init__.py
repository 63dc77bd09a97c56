"""CPU oracle for the tensor-field-convolution hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (pure PyTorch, fp64 or fp32, autograd backward) of
the arithmetic of eigenn's `PointConvMessagePassing` layer and its per-edge encoders.
Nothing in the shipped product (`matten_amd/`) imports it: only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may, and there only
as the checker / the timed baseline - never as the thing shipped.

PARITY UNPINNED.  The reference (`/root/reference/eigenn`) is pure Python whose
arithmetic lives in third-party packages that are absent from this image and from
`/root/reference` (e3nn - unpinned in the reference's setup.py:34, era ~0.4.x;
torch_scatter; ase).  The reference's own tests hold no golden vector for this path
(tests/nn/test_convolution.py:8-21 only constructs a module).  The oracle therefore
restates e3nn's published algorithm (`oracle/e3ref.py`) and the reference's composition
of it (`oracle/eigenn_ref.py`, each function citing the reference file:line it
follows), and is defended by first-principles tests: known Clebsch-Gordan values
checked against sympy, rotation/inversion equivariance in fp64, gradcheck, and the
structural irreps tables of SURVEY.md App. B.
"""
