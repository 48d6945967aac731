"""Drop-in for the reference `model/RAFT/raft.py`: `RAFT(args)(image1, image2, iters=12, test_mode=True) -> (flow_low, flow_up)`."""
import importlib

RAFT = importlib.import_module("zero-tig_amd.network").RAFT
