"""Input side of the path: batch collation and host->device transport.

Mirrors the names of pretrain_src/data: the six `*_collate` functions (r2r_tasks.py) and `PrefetchLoader` /
`move_to_cuda` (loader.py:77-124).  Not here: the HDF5 / jsonl readers (r2r_data.py) -- h5py and jsonlines are not in
this image."""
from .collate import (PackedBatch, itm_collate, mlm_collate, mrc_collate, sap_collate, sar_collate, sprel_collate,  # noqa: F401
                      COLLATE)
from .loader import PrefetchLoader, move_to_cuda  # noqa: F401
