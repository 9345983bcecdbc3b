"""A/B of attention kernels at model shapes (development aid): python tools/attn_ab.py"""
import os
import sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import attn_bench as T
for n in (3457, 6913, 11521, 23041):
    T.attn(2, n)
