"""Exercise bench.py's world-size-2 code path on a box with ONE GPU: two processes share cuda:0,
gradients are all-reduced through gloo (RCCL refuses two ranks on one device) and the recurrence
runs through the streaming kernels (two persistent launches from different processes cannot be
co-resident).  Checks plumbing only - the numbers mean nothing."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    procs = []
    for rank in (0, 1):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK='0', WORLD_SIZE='2',
                   MASTER_ADDR='127.0.0.1', MASTER_PORT='29611', CTCASR_DIST_BACKEND='gloo',
                   CTCASR_RNN_MODE='stream', CTCASR_BENCH_SHARE_GPU='1')
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus',
                                       '2', '--steps', '3', '--warmup', '2', '--workload', 'c2']
                                      + sys.argv[1:],
                                      env=env))
    codes = [p.wait(timeout=600) for p in procs]
    print('exit codes', codes)
    sys.exit(max(codes))


if __name__ == '__main__':
    main()
