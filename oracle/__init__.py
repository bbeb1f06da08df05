"""CPU restatement of the reference path: TEST INFRASTRUCTURE ONLY (imported by tests/, __graft_entry__.smoke(), bench.py cpu_baseline)."""
