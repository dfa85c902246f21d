#!/usr/bin/env python3
"""Start a script as N ranks the way torch.distributed.run does -- one process per rank with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT in its environment -- without importing torch.  Prints rank 0's stdout.

    python tools/launch_ranks.py N script.py [args...]
"""
import os
import socket
import subprocess
import sys


def main():
    n, script, args = int(sys.argv[1]), sys.argv[2], sys.argv[3:]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, script] + args, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = p.wait() or rc
    sys.stdout.write(out.decode())
    sys.exit(rc)


if __name__ == "__main__":
    main()
