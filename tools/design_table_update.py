#!/usr/bin/env python3
"""replace the text between <!-- r06-table-begin --> and <!-- r06-table-end --> in DESIGN.md by tools/design_numbers.py's output"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
r = sys.argv[1] if len(sys.argv) > 1 else "r06"
p = os.path.join(root, "DESIGN.md")
s = open(p).read()
a, b = "<!-- %s-table-begin -->\n" % r, "<!-- %s-table-end -->" % r
i, j = s.index(a) + len(a), s.index(b)
t = subprocess.run([sys.executable, os.path.join(root, "tools", "design_numbers.py"), r], capture_output=True, text=True, check=True).stdout
open(p, "w").write(s[:i] + t + s[j:])
print("updated", len(t), "bytes")
