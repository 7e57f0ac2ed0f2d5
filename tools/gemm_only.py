import os, sys
sys.argv=[sys.argv[0]]
sys.path.insert(0, "tools")
import time_tfm
time_tfm.gemm_bench()
