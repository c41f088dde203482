"""TEST INFRASTRUCTURE ONLY.  CPU restatements (plain PyTorch / numpy) of the reference path, used as the
parity checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under
multimodal_unlearning_amd/ may import this package."""
