"""A/B of attention kernels at model shapes (development aid): python tools_attn_ab.py"""
import sys, torch
sys.path.insert(0, ".")
import tools_attn_bench as T
for n in (3457, 6913, 11521, 23041):
    T.attn(2, n)
