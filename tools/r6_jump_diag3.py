"""research (round 6): holdout draws 9 / 11 with the strongly coupled pairs in the preconditioner"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from r6_jump_diag import run
for d in (9, 11):
    run(d, reps=5)
    run(d, verbose=1)
    run(d, viscosity_pair_correction=-1)
    run(d, viscosity_massless_polish=-1)
for d in (0, 6, 20, 30) if False else (0, 6):
    run(d); run(d, viscosity_pair_correction=-1)
