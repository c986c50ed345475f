"""CPU oracle for the MedTutor local dense-retrieval path.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package imports this
directory; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may use it, and there only as the checker or the reported CPU baseline.
"""
