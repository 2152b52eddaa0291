"""stdin: output of tools/gpu_legs.py lockstep... (stdout + stderr) -> one line per run + the phase timing lines"""
import json,sys
for line in sys.stdin:
    line=line.strip()
    if line.startswith("{"):
        for r in json.loads(line)["vo_lockstep"]["runs"]: print(r)
    elif "lockstep timing" in line: print(line)
