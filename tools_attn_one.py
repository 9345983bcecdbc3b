import sys, torch
sys.path.insert(0, ".")
import tools_attn_bench as T
T.attn(2, int(sys.argv[1]) if len(sys.argv) > 1 else 11521)
