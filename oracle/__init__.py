"""CPU oracle for the SHAPY hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; shapy_amd/ (the product) never does.
"""
