"""Input side of the path, behind the names of pretrain_src/data:

* `r2r_data`  trajectory (jsonl) / view-feature (HDF5, npz, npy) readers, `MultiStepNavData.get_input` (r2r_data.py)
* `r2r_tasks` the six task datasets `MlmDataset` ... `SprelDataset` and `random_word` (r2r_tasks.py)
* `collate`   the six `*_collate` functions (r2r_tasks.py) -- packed transport, padding on the device
* `loader`    `MetaLoader`, `PrefetchLoader`, `move_to_cuda`, `build_dataloader` (loader.py)"""
from .collate import (PackedBatch, itm_collate, mlm_collate, mrc_collate, sap_collate, sar_collate, sprel_collate,  # noqa: F401
                      COLLATE)
from .loader import MetaLoader, PrefetchLoader, build_dataloader, move_to_cuda  # noqa: F401
from .r2r_data import MultiStepNavData, ViewFeatureStore, read_jsonl  # noqa: F401
from .r2r_tasks import ItmDataset, MlmDataset, MrcDataset, SapDataset, SarDataset, SprelDataset, random_word  # noqa: F401
