"""oracle/ — TEST INFRASTRUCTURE, not product code.

CPU restatement (fp32 PyTorch) of the reference's GS-LoRA forgetting step plus the
harness that imports the real reference in the build container to pin it. Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import from here.
"""
