"""CPU oracle -- TEST INFRASTRUCTURE, not product code.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (plslam_amd) never does.
"""
