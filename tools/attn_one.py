import os
import sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import attn_bench as T
T.attn(2, int(sys.argv[1]) if len(sys.argv) > 1 else 11521)
