#!/usr/bin/env python3
"""Top entries of a cProfile dump (host-side cost of a bench workload):  python tools/prof_top.py file.prof [n]"""
import pstats
import sys

p = pstats.Stats(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 45
p.sort_stats("cumulative").print_stats(n)
p.sort_stats("tottime").print_stats(n)
