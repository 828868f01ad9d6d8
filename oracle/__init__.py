"""CPU oracle for the PixelPick hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (pixelpick_amd/) never does.
"""
