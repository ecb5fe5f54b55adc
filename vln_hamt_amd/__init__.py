"""vln_hamt_amd: MI355X-native implementation of the HAMT hot path (model forward/backward).

(`vln-hamt_amd` at the repository root is a symbolic link to this directory.)  Layout:

* ``csrc/``            hand-written HIP kernels for gfx950 + the C-ABI (``include/hamt.h``)
* ``_lib.py``          ctypes loader for ``libhamt_hip.so`` (fails loudly when missing)
* ``ops.py``           autograd Functions over the C-ABI
* ``model/``           mirror of the reference's ``pretrain_src/model`` class surface
* ``models/``          mirror of ``finetune_src/models``
* ``optim/``           HF-style AdamW / schedule / name-based decay groups on the flat arenas
* ``parallel.py``      data-parallel gradient reduction over RCCL
* ``synth.py``         synthetic batches with the reference's collate conventions
"""
__version__ = "0.1.0"
