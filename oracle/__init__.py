"""CPU oracle for the Building-GAN message-passing hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain PyTorch/numpy restatement, on the CPU, of the reference algorithm
(/root/reference/Model/models.py, util.py, util_graph.py, Model/losses.py) together with the
documented behaviour of the two un-vendored third-party packages the reference calls into
(torch_geometric 1.6.2, README.md:48, and the torch_scatter 2.0.x line it requires).  Only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
it; the product package ``building-gan_amd`` never does.

PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this path
(SURVEY.md section 4), and its own modules cannot be imported here because torch_geometric /
torch_scatter are not installed (ordinary ModuleNotFoundError, no permission was denied).  The
oracle is therefore checked only against hand-computed known-answer cases for each upstream
semantic (tests/test_oracle_kat.py) and against dense fp64 formulations, not against outputs of
the reference itself.
"""
