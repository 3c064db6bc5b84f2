"""Core-slot scheduler for the oracle worker processes of the full-size parity checks (test infrastructure).

Every worker runs a FIXED number of torch threads (16: the oracle's fp32 figures are bit-identical run to run at a fixed thread
count, SURVEY.md section 8c, and the measured bars of the full-size checks were taken at 16) on a slot of 16 logical CPUs of its
own; jobs beyond the number of slots wait in submission order.  One daemon thread starts / reaps the processes."""
from __future__ import annotations

import os
import subprocess
import threading
import time
from typing import List, Optional

SLOT = 16


class Job:
    def __init__(self, cmd, env, cwd):
        self.cmd, self.env, self.cwd = cmd, env, cwd
        self.proc: Optional[subprocess.Popen] = None
        self.done = threading.Event()
        self.rc: Optional[int] = None
        self.t_submit = time.perf_counter()
        self.t_start = self.t_end = None

    def wait(self, timeout: float) -> int:
        if not self.done.wait(timeout):
            raise TimeoutError(f"oracle worker did not finish within {timeout:.0f} s: {' '.join(self.cmd[-5:])}")
        return int(self.rc)


class Scheduler:
    def __init__(self, cores: List[int], slot: int = SLOT, slots: Optional[List[List[int]]] = None):
        self.slots = slots or [cores[i:i + slot] for i in range(0, len(cores) - slot + 1, slot)] or [list(cores)]
        self.free = list(range(len(self.slots)))
        self.queue: List[Job] = []
        self.running = []  # (job, slot index)
        self.lock = threading.Lock()
        self.thread: Optional[threading.Thread] = None
        self.stop = False

    def submit(self, cmd, env, cwd) -> Job:
        job = Job(cmd, env, cwd)
        with self.lock:
            self.queue.append(job)
            if self.thread is None or not self.thread.is_alive():
                self.thread = threading.Thread(target=self._loop, daemon=True)
                self.thread.start()
        return job

    def _loop(self):
        while True:
            with self.lock:
                for job, si in list(self.running):
                    rc = job.proc.poll()
                    if rc is not None:
                        job.rc, job.t_end = rc, time.perf_counter()
                        job.done.set()
                        self.running.remove((job, si))
                        self.free.append(si)
                while self.queue and self.free and not self.stop:
                    job, si = self.queue.pop(0), self.free.pop(0)
                    cores = self.slots[si]
                    env = dict(job.env, HARL_ORACLE_CORES=",".join(str(c) for c in cores), OMP_NUM_THREADS=str(len(cores)),
                               HARL_ORACLE_THREADS=str(len(cores)))
                    job.proc = subprocess.Popen(job.cmd, cwd=job.cwd, env=env)
                    job.t_start = time.perf_counter()
                    self.running.append((job, si))
                if not self.queue and not self.running:
                    self.thread = None
                    return
            time.sleep(0.25)

    def shutdown(self):
        with self.lock:
            self.stop = True
            for job in self.queue:
                job.rc = -1
                job.done.set()
            self.queue.clear()
            for job, _ in self.running:
                if job.proc.poll() is None:
                    job.proc.kill()


def physical_cores(avail: List[int]) -> List[List[int]]:
    """The logical CPUs of ``avail`` grouped by physical core (sysfs thread_siblings_list), in core order; one group per CPU when
    the topology cannot be read."""
    seen, groups = set(), []
    for c in sorted(avail):
        if c in seen:
            continue
        sib = [c]
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as f:
                txt = f.read().strip()
            sib = []
            for part in txt.split(","):
                lo, _, hi = part.partition("-")
                sib += list(range(int(lo), int(hi or lo) + 1))
            sib = [x for x in sib if x in set(avail)] or [c]
        except (OSError, ValueError):
            sib = [c]
        seen.update(sib)
        groups.append(sorted(sib))
    return groups


def partition(avail: List[int], main_logical: int, slot: int = SLOT):
    """(logical CPUs of the test process, list of worker slots): whole physical cores on either side -- the test process gets
    the first cores up to ``main_logical`` logical CPUs, every worker slot ``slot`` logical CPUs = slot / 2 cores with both of
    their hardware threads (no physical core is shared between two jobs or with the test process)."""
    cores = physical_cores(avail)
    main, k = [], 0
    while k < len(cores) and len(main) < main_logical:
        main += cores[k]
        k += 1
    slots, cur = [], []
    for grp in cores[k:]:
        cur += grp
        if len(cur) >= slot:
            slots.append(cur[:slot])
            cur = cur[slot:]
    return main, slots


ACTIVE: Optional[Scheduler] = None
