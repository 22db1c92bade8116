#!/usr/bin/env python3
"""DESIGN.md stays auditable: at most 400 lines of at most 120 characters.  usage: python tools/check_docs.py"""
import os
import sys

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "DESIGN.md")
lines = open(path, encoding="utf-8").read().split("\n")
bad = [(i, len(l)) for i, l in enumerate(lines, 1) if len(l) > 120]
print("DESIGN.md: %d lines, %d longer than 120 characters" % (len(lines), len(bad)))
for i, n in bad:
    print("  line %d: %d" % (i, n))
sys.exit(1 if bad or len(lines) > 400 else 0)
