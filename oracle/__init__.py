"""oracle — CPU restatements of the reference's retrieval hot path.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; nothing under
hair-centric-image-retrieval_amd/ does.  It never runs on the product path.
"""
