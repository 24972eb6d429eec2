import os
from itertools import chain, groupby
from typing import Any, List


def remove_duplicated_consecutive_elems_from_list(the_list: List[Any]) -> List[Any]:
    return [key for key, _ in groupby(the_list)]


def flatten_list(list_of_lists: List[List[Any]]) -> List[Any]:
    return list(chain.from_iterable(list_of_lists))


def equal_msas(msa_1, msa_2) -> bool:
    """Alignments are equal when their FASTA renderings are (reference utils/misc.py:16-22)."""
    return format(msa_1, "fasta") == format(msa_2, "fasta")


def should_output_debug_graphs() -> bool:
    return "make_prg_output_debug_graphs" in os.environ


def effective_cpus() -> int:
    """CPUs this process may actually use: the affinity mask, cut by the container's CPU quota (cgroup v2 cpu.max or v1
    cfs quota).  os.cpu_count() reports the machine (256 on the MI355X boxes) even where the quota is 16 CPUs."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return max(1, n)
