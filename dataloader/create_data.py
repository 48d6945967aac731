"""`CreateDataset(args, task)` with the reference's dataset names (dataloader/create_data.py:3-18).  Host-side file I/O only."""
from .multi_read_data import DefaultDataset, DidDataloader, FolderSequenceDataset, RLVDataLoader, SDSDDataloader  # noqa: F401


def CreateDataset(args, task):
    name = args.dataset
    if name in ("lowlight_dataset", "RLV", "BVI-RLV"):
        ds = RLVDataLoader()
    elif name in ("DID", "DID_1080"):
        ds = DidDataloader()
    elif name in ("SDSD", "3_SDSD"):
        ds = SDSDDataloader()
    else:
        ds = DefaultDataset()                 # e.g. `underwater` (is_WB path, model.py:94)
    print("dataset [%s] was created" % ds.name())
    ds.initialize(args, task)
    return ds
