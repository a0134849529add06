"""Host planner (tq_api.cpp: build_group_chunks) without a GPU: tools/planbench/plan_check.cpp
includes the planner's translation unit, plans synthetic launch groups (candidate unions with a
run of tiles per leader, AND-style groups, window unions) and checks that the chunks tile the
tile range exactly once, that every chunk is launched exactly once and that the per-query chunk
ranges are consistent — with the calling thread alone and with the planner's worker threads."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def plan_check(tmp_path_factory):
    from tantivy_amd import build as B

    B.build()  # the kernel objects the planner's translation unit links against
    out = tmp_path_factory.mktemp("plan") / "plan_check"
    obj = str(out) + ".o"
    src = os.path.join(ROOT, "tools", "planbench", "plan_check.cpp")
    subprocess.check_call([HIPCC, "-O1", "-std=c++17", "-Wno-unused-function", "-fPIC", "-c", src, "-o", obj],
                          cwd=str(out.parent))
    objs = [os.path.join(B.OBJ_DIR, os.path.basename(s) + ".o") for s in B.SOURCES if os.sep + "csrc" + os.sep in s]
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-o", str(out), obj] + objs + ["-ldl", "-lpthread"],
                          cwd=str(out.parent))
    return str(out)


@pytest.mark.parametrize("threads", [1, 4])
@pytest.mark.parametrize("chunks", [512, 131072])
def test_chunk_tables_cover_every_tile_once(plan_check, threads, chunks):
    for seed in (1, 2, 3):
        env = dict(os.environ, TQ_PLAN_THREADS=str(threads), TQ_CHUNKS=str(chunks), TQ_PLAN_PAR_MIN="1")
        r = subprocess.run([plan_check, str(seed)], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr + r.stdout
        assert r.stdout.count("ok") == 3, r.stdout


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_share_plan_invariants(plan_check, seed):
    """build_share_plan (the shared-union launch's planner): leads, lead records, tasks and result
    regions against a host-side restatement of their definitions."""
    for env_extra in ({}, {"TQ_US_TASK_COST": "256", "TQ_US_GROUP": "8"}, {"TQ_US_TASK_BLOCKS": "16"},
                      {"TQ_US_LIST_MB": "1"}):  # (a result-list budget that forces longer tasks)
        r = subprocess.run([plan_check, str(seed), "share"], env=dict(os.environ, **env_extra),
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr + r.stdout
        assert "ok" in r.stdout


@pytest.mark.parametrize("threads", [2, 4, 8])
def test_planner_thread_pool_runs_every_slab_once(plan_check, threads):
    """parallel_slabs / PlanPool: four host threads issuing 3000 jobs each (1..23 slabs) against the
    shared pool — every slab exactly once, no lost or doubled work, no deadlock."""
    r = subprocess.run([plan_check, "3", "pool"], env=dict(os.environ, TQ_PLAN_THREADS=str(threads)),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "pool: 4 callers x 3000 jobs ok" in r.stdout


@pytest.mark.parametrize("seed", range(6))
def test_dense_union_plan_invariants(plan_check, seed):
    """build_dense_plan (the doc-major union launch, tq_xunion.hip): one row per (list, weight) pair,
    bitmap rows first, every query's row bytes lead back to its lists, padding = the all-zero row,
    tasks cover every tile once, result lists disjoint."""
    r = subprocess.run([plan_check, str(seed), "dense"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "dense:" in r.stdout and "ok" in r.stdout


@pytest.mark.parametrize("seed", range(6))
def test_ashare_plan_invariants(plan_check, seed):
    """build_ashare_plan (the shared-intersection launch, tq_ashare.hip): one lead per query sorted by
    (leader, cache, mask), lead records against their queries, the tasks of every lead group tile the
    leader's blocks once, doc-slice launch order, disjoint result lists — also with small groups,
    long tasks and a result-list budget that forces longer tasks."""
    for env_extra in ({}, {"TQ_AS_GROUP": "5", "TQ_AS_TASK_PAIRS": "64"}, {"TQ_AS_TASK_BLOCKS": "7"},
                      {"TQ_AS_LIST_MB": "1"}, {"TQ_AS_DEDUPE": "0"}):  # (0: the comparison sort, identical queries as twins)
        r = subprocess.run([plan_check, str(seed), "ashare"], env=dict(os.environ, **env_extra),
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr + r.stdout
        assert "ashare:" in r.stdout and "ok" in r.stdout


@pytest.mark.parametrize("seed", range(6))
def test_boolean_share_plan_invariants(plan_check, seed):
    """build_ashare_plan with boolean leads (TQ_MODE_BOOL through the shared launch): one lead per (query,
    list of its lead set), tasks tile every lead's list once, result regions hold k entries per (task, lead)
    pair — and the filter stage is SAFE: over random doc-matrix words (exact columns, signature bits with
    false positives) no lead's masks drop a doc the query matches, no doc's bound is below what it can
    score, no list holding the doc is marked never-probed."""
    for env_extra in ({}, {"TQ_AS_GROUP": "3", "TQ_BS_TASK_PAIRS": "32"}, {"TQ_AS_LIST_MB": "1"}, {"TQ_AS_TASK_BLOCKS": "3"},
                      {"TQ_AS_DEDUPE": "0"}):
        r = subprocess.run([plan_check, str(seed), "bshare"], env=dict(os.environ, **env_extra),
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr + r.stdout
        assert "bshare:" in r.stdout and "ok" in r.stdout


@pytest.mark.parametrize("seed", range(4))
def test_count_expressions_match_boolean_semantics(plan_check, seed):
    """tq_count.cpp::count_expression (a query as a bitwise expression over bitmap words) + the loop body of
    count_bitmap_kernel restated on the host: 4 000 random AND / OR / boolean queries per seed (nested unions,
    absent terms, minimum_number_should_match, lists with and without a bitmap of their own) over random doc
    sets count exactly BooleanWeight's doc set; only m-of-n Should queries are handed to the scan."""
    r = subprocess.run([plan_check, str(seed), "count"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "count:" in r.stdout and "ok" in r.stdout


def test_eight_planners_at_once(tmp_path_factory):
    """eight ranks of a node plan their batches at the same time on the node's few
    granted CPUs.  Eight concurrent processes (one per CPU where there are eight) each plan the headline
    batch's shared-intersection tables 40 times on the calling thread (TQ_PLAN_THREADS defaults to 1: no
    helper threads to oversubscribe the cores with): the median under that load must stay within 2.5 x
    the uncontended median — planning must not collapse when every rank plans at once."""
    import re
    import shutil

    from tantivy_amd import build as B

    B.build()
    out = tmp_path_factory.mktemp("planb") / "plan_bench"
    obj = str(out) + ".o"
    src = os.path.join(ROOT, "tools", "planbench", "plan_bench.cpp")
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "-Wno-unused-function", "-fPIC", "-c", src, "-o", obj],
                          cwd=str(out.parent))
    objs = [os.path.join(B.OBJ_DIR, os.path.basename(s) + ".o") for s in B.SOURCES if os.sep + "csrc" + os.sep in s]
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-o", str(out), obj] + objs + ["-ldl", "-lpthread"],
                          cwd=str(out.parent))

    def median_of(stdout):
        m = re.search(r"median ([0-9.]+) ms", stdout)
        assert m, stdout
        return float(m.group(1))

    env = dict(os.environ, TQ_PLAN_THREADS="1")
    alone = median_of(subprocess.run([str(out), "10000", "40", "ashare"], env=env, capture_output=True, text=True,
                                     timeout=300).stdout)
    cpus = sorted(os.sched_getaffinity(0))
    pin = shutil.which("taskset") is not None
    procs = []
    for i in range(8):
        cmd = [str(out), "10000", "40", "ashare"]
        if pin:
            cmd = ["taskset", "-c", str(cpus[i % len(cpus)])] + cmd
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True))
    meds = [median_of(p.communicate(timeout=600)[0]) for p in procs]
    assert all(p.returncode == 0 for p in procs)
    slack = 2.5 * max(1.0, 8.0 / len(cpus))  # (fewer than eight CPUs: the processes share cores)
    assert max(meds) <= slack * alone + 0.5, (alone, meds)


def test_planners_under_thread_sanitizer(tmp_path_factory):
    """the planner translation units and the thread pool built with
    -fsanitize=thread, every check of plan_check run with four planner threads: no data race report."""
    from tantivy_amd import build as B

    B.build()
    d = tmp_path_factory.mktemp("tsan")
    csrc = os.path.join(ROOT, "tantivy_amd", "csrc")
    units = ["tq_plan_chunks.cpp", "tq_plan_share.cpp", "tq_plan_misc.cpp"]
    tsan_objs = []
    for u in units + [os.path.join(ROOT, "tools", "planbench", "plan_check.cpp")]:
        src = u if os.path.isabs(u) else os.path.join(csrc, u)
        obj = str(d / (os.path.basename(src) + ".o"))
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fPIC",
                            "-Wno-unused-function", "-fsanitize=thread", "-c", src, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            pytest.skip("no ThreadSanitizer build here: " + r.stderr[-300:])
        tsan_objs.append(obj)
    others = [os.path.join(B.OBJ_DIR, os.path.basename(s) + ".o") for s in B.SOURCES
              if os.sep + "csrc" + os.sep in s and os.path.basename(s) not in units]
    exe = str(d / "plan_check_tsan")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-fsanitize=thread", "-o", exe] + tsan_objs + others +
                       ["-ldl", "-lpthread"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("ThreadSanitizer runtime not linkable here: " + r.stderr[-300:])
    env = dict(os.environ, TQ_PLAN_THREADS="4", TQ_PLAN_PAR_MIN="1", TQ_CHUNKS="512")
    for args in (["1"], ["2", "share"], ["3", "ashare"], ["2", "bshare"], ["3", "pool"], ["1", "dense"]):
        r = subprocess.run([exe] + args, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "ThreadSanitizer" not in r.stdout + r.stderr, (args, (r.stdout + r.stderr)[-2000:])
