"""Pre-flight of the IPC halo transport in a CHILD process (bench.py, round 4).

The transport maps other processes' device memory (hipIpcOpenMemHandle) and polls flag words other devices write: on a node
where any of that does not work, the failure mode seen so far is a HOST call that never returns (gpurun_out/r4t) -- nothing a
process can recover from.  So before a job lets the transport into its own process, every rank starts this module as a child
(no torch, a few seconds): launcher bootstrap with YASK_HIP_TRANSPORT=ipc on its own ports, a small grid on the job's rank grid,
three steps with the default schedule and three with the pipelined half-exchanges, exit code 0.  The parent waits with a
time-out and kills a child that hangs; the ranks then agree (bench.py) whether the transport may be a candidate at all.

    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the launcher;  python -m yask_amd.ipc_preflight STENCIL RX RY RZ"""
import os
import sys


def main():
    stencil, grid = sys.argv[1], [int(x) for x in sys.argv[2:5]]
    os.environ["YASK_HIP_TRANSPORT"] = "ipc"
    os.environ.setdefault("YASK_HIP_WAIT_TIMEOUT_S", "5")
    from yask_amd import yk_factory
    from yask_amd.kernel import yk_env
    yk_env.disable_debug_output()
    fac = yk_factory(stencil)
    env = fac.new_env()
    env.init_from_launcher()             # binds the device (LOCAL_RANK), connects the control mesh, maps the mailboxes
    world = env.get_num_ranks()
    assert world == grid[0] * grid[1] * grid[2], (world, grid)
    env.transport_loopback(1 << 16)
    assert env.sum_over_ranks(1) == world
    for opts in ("-overlap_comms -hip_planned_launch -no-hip_halves", "-overlap_comms -hip_halves"):
        soln = fac.new_solution(env)
        soln.set_overall_domain_size_vec([96 * grid[0], 48 * grid[1], 64 * grid[2]])
        soln.set_num_ranks_vec(grid)
        assert soln.apply_command_line_options(opts) == ""
        soln.prepare_solution()
        for k, v in enumerate(soln.get_vars()):
            v.set_elements_hash(1.0 + 0.25 * k, 0.1, hash_id=k)
        soln.run_solution(0, 2)
        soln.end_solution()
        del soln
    env.global_barrier()
    print(f"ipc preflight: rank {env.get_rank_index()} of {world} ok", flush=True)
    os._exit(0)                          # (no destructors: the verdict is in, the parent is waiting)


if __name__ == "__main__":
    main()
