"""The rank launcher behind `python bench.py --gpus N` (turbo-metrics_amd/launch.py), on CPU: N children with the torchrun
environment, rank 0's stdout forwarded, failures propagated, a dead rank does not leave the others hanging."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(body, n, tmp_path, timeout_s=None):
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent(body))
    drv = tmp_path / "drv.py"
    drv.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        from tm_pkg import tm
        assert "torch" not in sys.modules  # the launcher must not pull in torch (nothing may touch the GPU before the ranks start)
        sys.exit(tm.launch.spawn_ranks([sys.executable, {str(script)!r}], {n}, timeout_s={timeout_s!r}))
    """))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    return subprocess.run([sys.executable, str(drv)], capture_output=True, text=True, timeout=120, env=env)


def test_ranks_get_the_torchrun_environment_and_rank0_stdout_is_forwarded(tmp_path):
    r = _run("""
        import os
        print("rank", os.environ["RANK"], os.environ["LOCAL_RANK"], os.environ["WORLD_SIZE"], os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]) > 0)
    """, 3, tmp_path)
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip() == "rank 0 0 3 127.0.0.1 True"  # ranks 1, 2 print to /dev/null: ONE line reaches the caller


def test_two_ranks_rendezvous_over_gloo(tmp_path):
    r = _run("""
        import os, torch, torch.distributed as dist
        dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
        t = torch.tensor([float(dist.get_rank() + 1)], dtype=torch.float64)
        dist.reduce(t, dst=0)
        if dist.get_rank() == 0: print("sum", t.item(), dist.get_world_size())
        dist.destroy_process_group()
    """, 2, tmp_path)
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip().splitlines()[-1] == "sum 3.0 2"  # (gloo itself prints a connection line on stdout)


def test_failing_rank_fails_the_launch_and_stops_the_others(tmp_path):
    r = _run("""
        import os, sys, time
        if os.environ["RANK"] == "1": sys.exit(7)
        time.sleep(60)  # would hang the launch if the dead rank went unnoticed
    """, 2, tmp_path)
    assert r.returncode == 7
    assert "a rank failed" in r.stderr


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=2 but --gpus 4" in r.stderr
    env["WORLD_SIZE"] = "2"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=2 but --gpus 1" in r.stderr


def test_cpulist_parsing():
    from tm_pkg import tm
    p = tm.launch.parse_cpulist
    assert p("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert p("5") == {5} and p("") == set()
    assert p("3-1") == set() and p("a-b") == set() and p("-4") == set()


def test_rank_thread_caps_follow_the_quota_share_and_keep_a_smaller_setting(tmp_path):
    """cap_rank_threads: before numpy / torch load, OMP / MKL thread counts = this rank's share of the CPUs the job may use; what
    torchrun already set lower (OMP_NUM_THREADS=1) stays"""
    body = textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {ROOT!r})
        import bench
        launch = bench.launch_module()  # launch.py on its own, the way a bench rank loads it
        assert "torch" not in sys.modules and "numpy" not in sys.modules
        cpus = launch.effective_cpus()
        assert 1 <= cpus <= (os.cpu_count() or 1)
        share = launch.cap_rank_threads(4)
        assert share == max(1, cpus // 4), (share, cpus)
        print(share, os.environ["OMP_NUM_THREADS"], os.environ["MKL_NUM_THREADS"])
    """)
    script = tmp_path / "caps.py"
    script.write_text(body)
    env = {k: v for k, v in os.environ.items() if k not in ("OMP_NUM_THREADS", "MKL_NUM_THREADS")}
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=60, env=env)
    assert r.returncode == 0, r.stderr
    share, omp, mkl = r.stdout.split()
    assert omp == share and mkl == share
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=60, env=dict(env, OMP_NUM_THREADS="1", MKL_NUM_THREADS="999"))
    assert r.returncode == 0, r.stderr
    share, omp, mkl = r.stdout.split()
    assert omp == "1" and mkl == share


def test_binding_to_a_numa_node_narrows_the_affinity_and_children_inherit_it(tmp_path):
    """bind_to_numa_node against a made-up sysfs tree: the node's CPUs that the process may use; unknown node, a node that covers
    everything, or TM_NUMA_BIND=0 change nothing"""
    body = textwrap.dedent(f"""
        import os, sys, subprocess
        sys.path.insert(0, {ROOT!r})
        from tm_pkg import tm
        have = sorted(os.sched_getaffinity(0))
        root = {str(tmp_path)!r}
        def node(n, cpus):
            os.makedirs(os.path.join(root, f"node{{n}}"), exist_ok=True)
            open(os.path.join(root, f"node{{n}}", "cpulist"), "w").write(",".join(str(c) for c in cpus) + "\\n")
        node(0, have)                      # covers every CPU we have: nothing to narrow
        node(1, have[:1] + [100000])       # one of ours + one that does not exist here
        node(2, [100001])                  # none of ours
        b = tm.launch.bind_to_numa_node
        assert b(-1, root) is None and b(7, root) is None and b(2, root) is None
        assert b(0, root) is None and sorted(os.sched_getaffinity(0)) == have
        os.environ["TM_NUMA_BIND"] = "0"
        assert b(1, root) is None and sorted(os.sched_getaffinity(0)) == have
        del os.environ["TM_NUMA_BIND"]
        if len(have) > 1:
            assert b(1, root) == 1 and sorted(os.sched_getaffinity(0)) == have[:1]
            out = subprocess.run([sys.executable, "-c", "import os; print(sorted(os.sched_getaffinity(0)))"], capture_output=True, text=True).stdout
            assert out.strip() == str(have[:1]), out
        print("ok")
    """)
    script = tmp_path / "bind.py"
    script.write_text(body)
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr


def test_pci_numa_node_from_sysfs(tmp_path):
    from tm_pkg import tm
    d = tmp_path / "0000:c1:00.0"
    d.mkdir()
    (d / "numa_node").write_text("1\n")
    e = tmp_path / "0001:05:00.0"
    e.mkdir()
    (e / "numa_node").write_text("-1\n")
    f = tm.launch.pci_numa_node
    assert f(0, 0xC1, 0, str(tmp_path)) == 1
    assert f(1, 5, 0, str(tmp_path)) == -1      # the kernel does not know
    assert f(0, 0xC2, 0, str(tmp_path)) == -1    # no such device
