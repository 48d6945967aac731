"""`CreateDataset(args, task)` with the reference's dataset names (dataloader/create_data.py:3-18).  Host-side file I/O only."""
from .multi_read_data import FolderSequenceDataset, RLVDataLoader


def CreateDataset(args, task):
    name = args.dataset
    if name in ("lowlight_dataset", "RLV", "BVI-RLV"):
        ds = RLVDataLoader()
    else:
        ds = FolderSequenceDataset()          # DID / SDSD / underwater style: <root>/<sequence>/<NNNN>.png
    print("dataset [%s] was created" % ds.name())
    ds.initialize(args, task)
    return ds
