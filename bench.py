#!/usr/bin/env python3
"""Benchmark of the hot path: Fisher scoring of a synthetic pool of 32^3 2-class patches.

Workload (BASELINE.json configs[2], the one the metric "patches/sec Fisher-scored (32^3,
2-class)" is quoted on): NET-C (patch-wise 3-D U-Net-style net with fc head),
pool of 100,000 synthetic patches resident in HBM (counter-based generator, seed 1004,
patch ids = global positions so shards are reproducible), weights He-normal seed 14.
One "step" = one Fisher-scoring pass over the whole pool: per patch p1, |p1-.5| top-B
candidates, g0, g1, A_i (8x8), tr(A_i), and the pool sum of A_i.

  python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU.  Under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` (the
driver's form: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) this process IS one rank.  From a bare shell
(`python bench.py --gpus N`, no WORLD_SIZE) it is the PARENT: before anything touches the GPU it starts N fresh rank
processes of itself (pool_shard.spawn_ranks), relays rank 0's JSON line and exits non-zero if any rank failed.

Default = weak scaling: every rank scores its own `--pool` (100,000) patches of an N*100,000 pool (config 4's sharding).
`--pool-global G` = strong scaling: ONE pool of G patches (configs[3]: 1,000,000) in contiguous blocks of ceil(G/N).
Either way the top-B merge and the 8x8 Fisher-sum all-reduce (RCCL) are inside the timed region.  ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

# Per-patch algorithmic work of NET-C at 32^3 (SURVEY.md §8d / BASELINE.md §4), 2 FLOP per MAC.
F_FWD = 439.88e6
F_BWD_DATA = 425.72e6
F_SURVEY = F_FWD + 2 * F_BWD_DATA       # the survey's figure: one backward per class
F_EXEC = F_FWD + F_BWD_DATA             # what this build executes: ONE backward serves both classes
PEAK_BF16_MFMA_TFLOPS = 2500.0          # MI355X_MICROARCH.md: dense bf16 MFMA peak
PEAK_HBM_TBPS = 8.0                     # MI355X_MICROARCH.md: HBM3E
SPLIT_PRODUCTS = 6                      # bf16 MFMA MACs issued per fp32-accurate MAC (igemm3.hip)
B_ALG = 4 * 32 ** 3 + 4 * (1 + 2 * 8 + 64)   # input + outputs per patch, bytes
BOOST_SCLK_MHZ = 2400.0                 # MI355X_MICROARCH.md: the clock the dense peaks are quoted at


class ClockSampler(object):
    """Engine clock and socket power of the GPU this rank computes on, read from its hwmon files (`freq1_input`, `power1_input`:
    world-readable, no tool, no GPU call) every 25 ms between start() and stop().  The scoring pass holds the board at its power
    cap; the clock it sustains there - not the boost clock the datasheet peaks are quoted at - is what its matrix pipes run at."""

    def __init__(self, torch, index):
        self.dir = None
        self.samples = []
        self._stop = None
        try:
            pr = torch.cuda.get_device_properties(index)
            bdf = '%04x:%02x:%02x.0' % (getattr(pr, 'pci_domain_id', 0), pr.pci_bus_id, pr.pci_device_id)
            import glob
            for d in glob.glob('/sys/class/drm/card*/device'):
                if os.path.basename(os.path.realpath(d)) == bdf:
                    hw = glob.glob(os.path.join(d, 'hwmon', 'hwmon*'))
                    if hw and os.path.exists(os.path.join(hw[0], 'freq1_input')):
                        self.dir = hw[0]
                        self.bdf = bdf
        except Exception:
            self.dir = None

    def _read(self, name):
        try:
            with open(os.path.join(self.dir, name)) as f:
                return float(f.read().strip())
        except Exception:
            return None

    def start(self):
        if self.dir is None:
            return
        import threading
        self.samples = []
        self._stop = threading.Event()

        def run():
            while not self._stop.is_set():
                f, w = self._read('freq1_input'), self._read('power1_input')
                if f is not None:
                    self.samples.append((f / 1e6, (w or 0.0) / 1e6))
                self._stop.wait(0.025)
        self._thread = threading.Thread(target=run, daemon=True)
        self._thread.start()

    def stop(self):
        if self._stop is None:
            return None
        self._stop.set()
        self._thread.join()
        self._stop = None
        if not self.samples:
            return None
        f = sorted(s[0] for s in self.samples)
        w = sorted(s[1] for s in self.samples)
        cap = self._read('power1_cap')
        return {'sclk_mhz_mean': sum(f) / len(f), 'sclk_mhz_median': f[len(f) // 2], 'sclk_mhz_min': f[0], 'sclk_mhz_max': f[-1],
                'power_w_mean': sum(w) / len(w), 'power_w_max': w[-1], 'power_cap_w': cap / 1e6 if cap else None,
                'boost_sclk_mhz': BOOST_SCLK_MHZ, 'samples': len(f), 'device': self.bdf,
                'source': 'hwmon freq1_input / power1_input of the device, every 25 ms over the timed region'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--pool', type=int, default=100000, help='patches per GPU (weak scaling)')
    ap.add_argument('--pool-global', type=int, default=0, help='strong scaling: one pool of this many patches over all GPUs')
    ap.add_argument('--netb-pool', type=int, default=16384, help='patches of the NET-B side measurement at N = 1 (0 = skip)')
    ap.add_argument('--batch', type=int, default=2047,
                    help='most patches per device pass.  2047 = the most the engines\' unsigned 32-bit tensor offsets address for 32^3 NET-C '
                         '(the library clamps a larger request to it).  fisher_device cuts a pool into an EVEN number of equal passes of at most '
                         'this size, so that the two scoring pipelines get the same work (DeviceModel.pass_cut; 100,000 patches: 50 passes '
                         'of 2000; +1.75 %% same-box against 48 x 2047 + 1744)')
    ap.add_argument('--topB', type=int, default=4096)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-accuracy', action='store_true', help='skip the accuracy passes behind the timed region (PMC / sweep runs: their fp64 kernels would dominate a short trace)')
    ap.add_argument('--backend', default='nccl', choices=('nccl', 'gloo'),
                    help='torch.distributed backend of the N > 1 exchanges (nccl = RCCL over xGMI; gloo: functional rehearsal)')
    ap.add_argument('--same-gpu', action='store_true',
                    help='functional rehearsal on a one-GPU box: every rank computes on cuda:0 (use with --backend gloo; RCCL '
                         'refuses two ranks on one device); throughput numbers of such a run mean nothing')
    ap.add_argument('--prof-every', type=int, default=8, help='--lanes 1 only: HIP-event timing on every k-th pass of the timed region')
    ap.add_argument('--lanes', type=int, default=2,
                    help='scoring pipelines of the timed region (device.DeviceModel.lanes): 2 = device passes alternate between two libalq '
                         'contexts so that one pass\'s tail runs beside the next one\'s first launches (outputs bit-identical to 1, '
                         'test_two_scoring_pipelines_are_bit_identical_to_one).  Per-launch durations then include co-residency, so the '
                         'roofline object always comes from a SEPARATE single-pipeline pass with HIP events on every launch')
    ap.add_argument('--roofline-passes', type=int, default=12, help='device passes of the separate single-pipeline roofline pass (an even number: they are then cut like the timed region\'s)')
    ap.add_argument('--cpu-sample', type=int, default=128, help='patches the CPU baseline scores (~15 s on 16 cores)')
    ap.add_argument('--cpu-all-cores', action='store_true',
                    help='also time ONE patch of the CPU baseline with os.cpu_count() threads (BASELINE.md 3 names that thread count; at '
                         'batch 1 it oversubscribes every op: 0.03 patches/s on the 256-thread box, ~35 s)')
    ap.add_argument('--config', type=int, default=2, choices=(0, 1, 2, 3, 4, 5),
                    help='BASELINE.json configs[i]: 0 NET-A entropy query over 1,000 patches, 1 NET-A Fisher scoring of 10,000, 2 NET-C Fisher '
                         'scoring of 100,000 per GPU (the metric\'s config, default), 3 ONE pool of 1,000,000 over the GPUs (= --pool-global '
                         '1000000), 4 the active-learning loop, 5 rounds over 200,000 patches; 5 (not a BASELINE.json config) the VOLUME-level `fi` query the '
                         'reference times (PW_AL.py:848-855): query_multimg over synthetic padded subjects through alq_gather_normalize, '
                         'the reference\'s own (25,25,1) x 2-modality NET-B')
    args = ap.parse_args()
    if args.config == 3 and args.pool_global == 0:
        args.pool_global = 1000000

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # Bare-shell launch: this process only starts the ranks (it never imports torch.cuda or touches HIP).
        import nnal_amd  # noqa: F401
        from nnal_amd import pool_shard
        rc, out0 = pool_shard.spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus)
        sys.stdout.write(out0)
        sys.stdout.flush()
        sys.exit(rc)

    import torch
    import torch.distributed as dist
    ws = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if ws != args.gpus:
        sys.exit('bench.py: WORLD_SIZE=%d but --gpus %d' % (ws, args.gpus))
    if ws > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.same_gpu:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=ws, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group('gloo', rank=rank, world_size=ws)

    import ctypes as C
    import nnal_amd  # noqa: F401
    from nnal_amd import device, pool_shard
    from nnal_amd._lib import check
    from nnal_amd import netspec   # network definition + seeded weight draw (product copy; not timed)

    sess = device.DeviceSession(local_rank)
    if args.config in (0, 1, 4, 5):
        line = small_config(args, sess, rank, ws) if args.config < 4 else (loop_config(args, sess, rank, ws) if args.config == 4 else volume_config(args, sess, rank, ws))
        if rank == 0:
            print(json.dumps(line))
        if ws > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    ld, sk = netspec.net_c()
    in_shape = (32, 32, 32, 1)
    pars = netspec.he_init(ld, in_shape, seed=14, skips=sk)
    model = device.DeviceModel(sess, ld, in_shape, sk, max_batch=args.batch)
    model.set_weights(pars)
    args.batch = model.max_batch          # what the library granted (alq_model_max_batch)
    model.lanes = max(1, args.lanes)

    strong = args.pool_global > 0
    n_global = args.pool_global if strong else args.pool * ws
    a0, b0 = pool_shard.shard_bounds(n_global, ws, rank)
    n_local = b0 - a0
    assert strong or n_local == args.pool
    comm = 'torch.distributed (world 1: identity)'
    if ws > 1 and args.backend != 'nccl':
        comm = 'torch.distributed gloo (functional rehearsal%s)' % (', all ranks on cuda:0' if args.same_gpu else '')
    elif ws > 1:
        # the Fisher-sum all-reduce goes through the C ABI (alq_allreduce_sum, the library's own RCCL communicator)
        try:
            pool_shard.attach_comm(sess)
            comm = 'alq_allreduce_sum (RCCL communicator of the libalq context)'
        except Exception as e:      # a transport choice, not a compute fallback: torch.distributed's RCCL group does the same sum
            comm = 'torch.distributed nccl (alq_comm_init failed: %s)' % (e,)
            print('[bench] rank %d: %s' % (rank, comm), file=sys.stderr, flush=True)
    epp = 32 ** 3
    x = sess.empty((n_local, epp), torch.float32)      # 13.1 GB per GPU at the default pool
    check(sess.lib.alq_synth_patches(sess.ctx, 1004, a0, n_local, epp, C.c_void_p(x.data_ptr())))
    torch.cuda.synchronize()

    def step():
        return pool_shard.score_pool(model, sess, x, n_global, args.topB, 1e-3)

    def note(msg):
        if rank == 0:
            print('[bench] ' + msg, file=sys.stderr, flush=True)

    note('pool of %d patches resident (%.1f GB); warm-up' % (n_local, n_local * epp * 4 / 1e9))
    for _ in range(args.warmup):
        step()
    note('timing %d step(s)' % args.steps)
    sess.prof_reset()
    # one pipeline: HIP events on our stream around the launches of every 8th pass of the timed region (on every launch the
    # event pairs themselves cost ~6 % of the step); two pipelines: none - a launch's event span would include whatever the
    # other pipeline runs beside it - and the roofline pass below instead
    in_region = model.lanes == 1 and not os.environ.get('ALQ_BENCH_NO_EVENTS')
    sess.prof_enable(args.prof_every if in_region else 0)
    sampler = ClockSampler(torch, local_rank)
    pool_shard.barrier()
    torch.cuda.synchronize()
    sampler.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    pool_shard.barrier()
    dt = time.perf_counter() - t0
    clocks = sampler.stop()
    sess.prof_enable(False)
    prof = sess.prof_read()
    dt = pool_shard.max_over_ranks(dt)
    lanes_timed = int(model.lanes)
    prof_patches = None
    clocks_roof = clocks if in_region else None
    if not in_region and not os.environ.get('ALQ_BENCH_NO_EVENTS'):
        # the roofline pass: the same launches on the first passes of the pool, ONE pipeline, HIP events on every launch
        # (outside the timed region; what rocprofv3 --kernel-trace shows for `--lanes 1`)
        model.lanes = 1
        # ... in passes of the timed region's size: an even number of them is cut into exactly those passes
        step_timed = model.pass_cut(n_local)[0]
        prof_patches = min(n_local, max(2, args.roofline_passes // 2 * 2) * step_timed)
        sess.prof_reset()
        sess.prof_enable(1)
        sampler.start()
        model.fisher_device(x, prof_patches, None, 1e-3, want=('p1', 'H', 'g0', 'g1', 'A', 'trace', 'Asum'))
        torch.cuda.synchronize()
        clocks_roof = sampler.stop()
        sess.prof_enable(False)
        prof = sess.prof_read()
        model.lanes = lanes_timed

    if rank == 0:
        ms_per_step = 1e3 * dt / args.steps
        value = n_global * args.steps / dt
        # dominant kernel = igemm4_kernel (conv / conv_transpose fwd + bwd-data on the bf16 matrix cores
        # with the 3-way operand split: 6 bf16 MFMA MACs per fp32-accurate MAC, so the matrix-core bound
        # on ALGORITHMIC fp32 flops is the dense bf16 peak / 6)
        # The launches whose input scale is known ahead of time (the conv under the head, forward and backward - the plane-sweep
        # kernels of csrc/c3d.hip - and every other backward launch) run on fp16 pairs: 3 fp16 MFMA MACs per fp32-accurate MAC,
        # bound = dense fp16 peak (= the bf16 one) / 3.  The bound of the mix is the flop-weighted harmonic mean of the two.
        f16 = prof.get('igemm_f16x2', {'ms': 0.0, 'flops': 0.0, 'launches': 0})
        bf_fl = prof['igemm3_fwd']['flops'] + prof['igemm3_bwd']['flops']
        ig_ms = prof['igemm3_fwd']['ms'] + prof['igemm3_bwd']['ms'] + f16['ms']
        ig_fl = bf_fl + f16['flops']
        ig_n = prof['igemm3_fwd']['launches'] + prof['igemm3_bwd']['launches'] + f16['launches']
        # patches behind the sampled launches: 12 igemm4 launches per device pass of `batch` patches (the pool divides evenly
        # at the default sizes; a ragged last pass would be counted at full size, so derive the count from the passes)
        passes_per_step = len(model.pass_cut(n_local)[1])
        sampled_passes = ig_n / 12.0
        if prof_patches is None:
            prof_patches = sampled_passes * (n_local / float(passes_per_step))
        achieved = ig_fl / (ig_ms * 1e-3) / 1e12 if ig_ms > 0 else 0.0
        peak_bf, peak_f16 = PEAK_BF16_MFMA_TFLOPS / SPLIT_PRODUCTS, PEAK_BF16_MFMA_TFLOPS / 3
        peak = ig_fl / (bf_fl / peak_bf + f16['flops'] / peak_f16) if ig_fl > 0 else peak_bf
        executed = (bf_fl * SPLIT_PRODUCTS + f16['flops'] * 3) / (ig_ms * 1e-3) / 1e12 if ig_ms > 0 else 0.0
        conv_ms = ig_ms + sum(prof[k]['ms'] for k in ('igemm_fwd', 'igemm_bwd', 'direct_conv'))
        conv_fl = ig_fl + sum(prof[k]['flops'] for k in ('igemm_fwd', 'igemm_bwd', 'direct_conv'))
        traffic, traffic_note = None, None   # PMC passes are separate runs (tools/run_pmc.sh -> profiles/pmc_traffic.json)
        tp = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        if os.path.exists(tp):
            try:
                tj = json.load(open(tp))
                # measured per launch at tj['batch'] patches; a launch moves bytes in proportion to its patches
                per_launch = prof_patches / max(ig_n / 12.0, 1.0)      # patches behind one of the launches `avg_launch_ms` averages
                traffic = tj.get('hbm_bytes_per_launch') * per_launch / float(tj.get('batch', args.batch))
                traffic_note = ('PMC 2*FETCH_SIZE + WRITE_SIZE per contraction launch, measured at %d patches per launch (%s), scaled to the '
                                '%.0f patches per launch of the launches timed here' % (tj.get('batch', args.batch), 'profiles/pmc_traffic.json', per_launch))
            except Exception:
                traffic = None
        line = {
            'metric': 'patches/sec Fisher-scored (32^3, 2-class)',
            'value': value, 'unit': 'patches/s', 'n_gpus': ws, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'strong' if strong else 'weak', 'vs_baseline': None,
            'dtype': 'f32 (fp32 operands split into 16-bit pieces on the 16-bit MFMA, fp32 accumulate: fp16 pairs - 22-23 bits kept per '
                     'operand, three products - in the conv under the head (forward + backward), in dec1\'s forward launch and in the other '
                     'backward launches; bf16 triples - all 24 bits, exact, six products - in the remaining forward launches; everything else '
                     'fp32 / fp64)',
            'data': 'synthetic',
            'config': {'workload': ('configs[3]: Fisher scoring, NET-C patch-wise 3-D U-Net (fc head), ONE pool of %d synthetic 32^3 '
                                    '2-class patches in contiguous blocks over the GPUs, random-init weights seed 14' % n_global) if strong else
                                   ('configs[2]: Fisher scoring, NET-C patch-wise 3-D U-Net (fc head), '
                                    '%d synthetic 32^3 2-class patches per GPU, random-init weights seed 14' % n_local),
                       'outputs_per_patch': 'p1, |p1-.5| (top-B keys), H, g0[8], g1[8], A[8x8], tr A stored; sum A over the pool',
                       'pool_global': n_global, 'pool_per_gpu': n_local, 'batch': model.max_batch, 'topB': args.topB,
                       'passes_per_step': len(model.pass_cut(n_local)[1]), 'patches_per_pass': model.pass_cut(n_local)[0],
                       # integers the driver can check: ranks of the torch.distributed group and of the RCCL communicator the libalq
                       # context owns (0 = none: the Fisher sum goes through torch.distributed or, at world 1, nowhere)
                       'dist_world_size': dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1,
                       'comm_world': int(getattr(sess, 'comm_world', 0) or 0),
                       'dist_backend': dist.get_backend() if (dist.is_available() and dist.is_initialized()) else None,
                       'head_conv_engine': {'forward_plane_sweep': int(sess.lib.alq_model_engine_info(model._m, 1)),
                                            'backward_plane_sweep': int(sess.lib.alq_model_engine_info(model._m, 2)),
                                            'flip_list_overflow': int(sess.lib.alq_model_engine_info(model._m, 5))},
                       'parallelism': 'pool sharded over %d GPU(s), top-B merge (all-gather) + 8x8 Fisher all-reduce: %s' % (ws, comm)},
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s',
                         'frac': achieved / peak, 'traffic': traffic, 'traffic_note': traffic_note,
                         # the ratio that does not reward redundant products: algorithmic contraction flops against the best
                         # fp32-faithful split known (3 products per MAC) on the dense 16-bit peak
                         'useful_frac': achieved / peak_f16,
                         'useful_frac_definition': 'algorithmic fp32 flops of the 12 contraction launches / their HIP-event time / (%.0f TFLOP/s / 3 '
                                                   'products): what a launch would reach at most if every one ran the 3-product fp16-pair split'
                                                   % PEAK_BF16_MFMA_TFLOPS,
                         'kernel': 'the 12 contraction launches of a pass: c3d_fwd_kernel / c3d_bwd_kernel (the conv under the head: 53 % of the '
                                   'flops, fp16 pairs), d3d_fwd / d3d_bwd (dec1), t3d_fwd / t3d_bwd (up2), t3d8_fwd / t3d8_bwd (up1), f3d_fwd (enc2 forward + '
                                   'pool2), e3d_bwd (enc2 backward + both pool backwards) and two igemm4_kernel launches (bott forward / backward); '
                                   'fp16 pairs in the backward launches and in the dec1 / enc2 forward launches, bf16 triples in the other forward ones',
                         'peak_note': 'algorithmic fp32 flops; peak = flop-weighted harmonic mean of %.0f dense 16-bit MFMA / %d '
                                      'products (bf16x3 launches, %.0f %% of the flops) and / 3 (the fp16x2 launches)'
                                      % (PEAK_BF16_MFMA_TFLOPS, SPLIT_PRODUCTS, 100.0 * bf_fl / max(ig_fl, 1.0)),
                         'frac_definition': 'executed 16-bit MFMA flops (algorithmic flops x 6 products in the bf16x3 launches, x 3 in '
                                            'the fp16x2 ones) / igemm4 time / %.0f TFLOP/s = achieved / peak above; recomputed from a '
                                            'rocprofv3 --kernel-trace --stats run of this command by tools/roofline_from_stats.py'
                                            % PEAK_BF16_MFMA_TFLOPS,
                         'executed_16bit_tflops': executed, 'peak_16bit_tflops': PEAK_BF16_MFMA_TFLOPS,
                         'igemm4_alg_flops_per_patch': {'bf16x3': bf_fl / max(prof_patches, 1), 'f16x2': f16['flops'] / max(prof_patches, 1)},
                         'hbm_frac': (traffic * ig_n / (ig_ms * 1e-3) / (PEAK_HBM_TBPS * 1e12)) if traffic and ig_ms > 0 else None,
                         'f16x2_launches': {'tflops': f16['flops'] / (f16['ms'] * 1e-3) / 1e12 if f16['ms'] > 0 else 0.0,
                                          'avg_launch_ms': f16['ms'] / max(f16['launches'], 1), 'bound_tflops': peak_f16},
                         'launches': ig_n, 'avg_launch_ms': ig_ms / max(ig_n, 1),
                         'timed': ('HIP events on every k-th pass of the timed region, k = %d' % args.prof_every) if in_region else
                                  ('a separate single-pipeline pass over the first %d pool patches behind the timed region, HIP events on '
                                   'every launch (the timed region ran %d pipelines: its launches overlap each other)' % (int(prof_patches or 0), lanes_timed)),
                         'all_conv_engines_tflops': conv_fl / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0,
                         'flops_per_patch_executed': F_EXEC, 'flops_per_patch_survey': F_SURVEY,
                         'whole_step_tflops_executed': F_EXEC * value / ws / 1e12,
                         'hbm_algorithmic_GBps': B_ALG * value / ws / 1e9,
                         'time_share_ms_sampled': {k: v['ms'] for k, v in prof.items()},
                         'time_share_note': 'event spans; the reduce / fc_small kernels of the backward pass run on the side stream '
                                            'BESIDE the igemm launches (their spans overlap those, they do not add up to the step)'},
            # engine clock / board power while the timed region ran: the pass sits at the board's power cap, below the boost clock
            # the peaks above are quoted at (`frac` stays against the datasheet peak)
            'clocks': clocks,
        }
        if clocks_roof:
            r = line['roofline']
            r['clocks'] = clocks_roof
            r['sclk_ratio'] = clocks_roof['sclk_mhz_mean'] / BOOST_SCLK_MHZ
            r['frac_at_sustained_sclk'] = r['frac'] / r['sclk_ratio']
            r['frac_at_sustained_sclk_note'] = ('frac against the 16-bit MFMA peak scaled to the mean engine clock measured while these launches '
                                                'ran (%.0f of %.0f MHz at %.0f W of a %.0f W cap): what the matrix pipes could deliver at that clock; '
                                                'not a datasheet figure' % (clocks_roof['sclk_mhz_mean'], BOOST_SCLK_MHZ, clocks_roof['power_w_mean'],
                                                                            clocks_roof.get('power_cap_w') or 0.0))
        note('GPU: %.1f patches/s' % value)
        # what the LAST pass of the timed region ran on (asked before the accuracy passes below, one of which is the exact-fp32 engine)
        info = sess.lib.alq_model_engine_info
        line['config']['head_conv_engine']['dec1_forward_fp16_pairs'] = int(info(model._m, 6))
        line['config']['engines'] = {'dec1_forward_plane_sweep': int(info(model._m, 10)), 'dec1_backward_plane_sweep': int(info(model._m, 11)),
                                     'enc2_forward_fused_with_pool': int(info(model._m, 12)),
                                     'conv_transpose_forward_row_sweep_launches': int(info(model._m, 7)),
                                     'conv_transpose_backward_row_sweep_launches': int(info(model._m, 8)),
                                     'enc2_backward_fused_with_pool_backwards': int(info(model._m, 9)),
                                     'head_conv_backward_k_steps': {7: 7, 8: 9, 4: 9}.get(int(info(model._m, 13)), 0),
                                     'scoring_pipelines': lanes_timed}
        if ws == 1 and not args.no_accuracy:
            line['accuracy'] = accuracy_vs_exact_fp32(sess, model, x, min(args.batch, n_local), n64=min(args.batch, n_local))
        if ws == 1 and args.netb_pool > 0:
            line['netb'] = netb_rate(sess, args.netb_pool, x, accuracy=not args.no_accuracy)
        if not args.no_cpu_baseline and ws == 1:      # reported at N = 1 only (the other ranks would sit in the barrier)
            note('timing the CPU baseline')
            line['cpu_baseline'] = cpu_baseline(x[:args.cpu_sample].cpu().numpy(), ld, sk, in_shape, pars, args.cpu_all_cores)
        print(json.dumps(line))
    if ws > 1:
        dist.barrier()
        dist.destroy_process_group()


def _conv_roofline(prof, note):
    """Roofline object from the HIP-event classes of the contraction engines, against the split that RAN: launches of the
    fp16-pair class execute 3 products per MAC (ceiling 2500 / 3), the bf16-triple engines (and the first-layer / fp32 fallback
    kernels, counted with them) 6 (ceiling 2500 / 6); `peak` = their flop-weighted harmonic mean, `useful_frac` = algorithmic
    flops against 2500 / 3 as in the main line."""
    bf_keys = ('igemm_fwd', 'igemm_bwd', 'igemm3_fwd', 'igemm3_bwd', 'direct_conv')
    f16 = prof.get('igemm_f16x2', {'ms': 0.0, 'flops': 0.0, 'launches': 0})
    bf_fl = sum(prof[k]['flops'] for k in bf_keys if k in prof)
    ms = sum(prof[k]['ms'] for k in bf_keys if k in prof) + f16['ms']
    fl = bf_fl + f16['flops']
    n = sum(prof[k]['launches'] for k in bf_keys if k in prof) + f16['launches']
    peak_bf, peak_f16 = PEAK_BF16_MFMA_TFLOPS / SPLIT_PRODUCTS, PEAK_BF16_MFMA_TFLOPS / 3
    peak = fl / (bf_fl / peak_bf + f16['flops'] / peak_f16) if fl > 0 else peak_bf
    ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    return {'bound': 'mfma', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak, 'useful_frac': ach / peak_f16, 'traffic': None,
            'kernel': 'all contraction launches (first-layer kernel, bf16x3 engines, fp16-pair launches, fp32 fallback engine)', 'launches': n,
            'avg_launch_ms': ms / max(n, 1),
            'peak_note': 'algorithmic fp32 flops; peak = flop-weighted harmonic mean of %.0f dense 16-bit MFMA / 6 products (bf16x3 and first-layer '
                         'launches, %.0f %% of the flops) and / 3 (fp16-pair launches); useful_frac = achieved / (%.0f / 3)'
                         % (PEAK_BF16_MFMA_TFLOPS, 100.0 * bf_fl / max(fl, 1.0), PEAK_BF16_MFMA_TFLOPS),
            'alg_flops': {'bf16x3': bf_fl, 'f16x2': f16['flops']},
            'note': note, 'time_share_ms_sampled': {k: v['ms'] for k, v in prof.items()}}


def small_config(args, sess, rank, ws):
    """configs[0]: NET-A entropy query over 1,000 patches (RandomState(1001), weights seed 11, k = 50; the
    reference's CNN_query(..., 'entropy'), PW_NNAL.py:51-65); configs[1]: NET-A Fisher scoring of 10,000 patches (RandomState(1002),
    weights seed 12, diag_load 1e-5).  One GPU (these pools do not shard meaningfully); launch-latency bound by construction."""
    import torch
    from nnal_amd import device, netspec, pool_shard
    ent = args.config == 0
    n = 1000 if ent else 10000
    ld = netspec.net_a()
    in_shape = (32, 32, 1)
    pars = netspec.he_init(ld, in_shape, seed=11 if ent else 12)
    model = device.DeviceModel(sess, ld, in_shape, (), max_batch=min(n, 10000))
    model.set_weights(pars)
    xs = np.random.RandomState(1001 if ent else 1002).randn(n, *in_shape).astype(np.float32)
    x = sess.to_device(xs.reshape(n, -1), torch.float32)
    k = 50 if ent else 500

    def step():
        if ent:
            post, _, _ = model.forward_device(x, n)
            return sess.uncertainty_filter(post[1].contiguous(), k)
        return pool_shard.score_pool(model, sess, x, n, k, 1e-5)['sel']
    for _ in range(max(args.warmup, 1)):
        step()
    sess.prof_reset()
    sess.prof_enable(0 if os.environ.get('ALQ_BENCH_NO_EVENTS') else 1)
    steps = max(args.steps, 20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        sel = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sess.prof_enable(False)
    prof = sess.prof_read()
    line = {'metric': 'patches/sec %s (32x32x1, 2-class)' % ('entropy-scored' if ent else 'Fisher-scored'), 'value': n * steps / dt, 'unit': 'patches/s',
            'n_gpus': 1, 'steps': steps, 'warmup': max(args.warmup, 1), 'ms_per_step': 1e3 * dt / steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32 (bf16x3 operand split on the bf16 MFMA, fp32 accumulate; fp32 / fp64 elsewhere)', 'data': 'synthetic',
            'config': {'workload': ('configs[0]: entropy query, NET-A 3-layer 2-D CNN, 1,000 synthetic 32x32x1 patches (RandomState 1001), weights seed 11, k = 50'
                                    if ent else 'configs[1]: Fisher scoring, NET-A 3-layer 2-D CNN, 10,000 synthetic 32x32x1 patches (RandomState 1002), '
                                    'weights seed 12, diag_load 1e-5, top-500'), 'pool_global': n, 'batch': model.max_batch},
            'roofline': _conv_roofline(prof, 'a %d-patch pool of a 9,154-parameter net: %d kernel launches of a few microseconds each per step - '
                                       'launch latency, not a roofline, bounds it' % (n, sum(v['launches'] for v in prof.values()) // max(steps, 1)))}
    if not args.no_cpu_baseline:
        from oracle import alpath
        from oracle.model import OracleModel, OracleSession
        torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 16)))
        om = OracleModel(ld, in_shape, pars)
        if ent:      # the reference's entropy branch: batched forward + argsort of |p - .5| (PW_NNAL.py:51-65)
            t0 = time.perf_counter()
            reps = 0
            while time.perf_counter() - t0 < 3.0:
                p = om.forward(xs)['posteriors'][1]
                q = np.argsort(np.abs(p.astype(np.float64) - .5))[:k]
                reps += 1
            dtc = time.perf_counter() - t0
            assert np.array_equal(np.sort(q), np.sort(sel.cpu().numpy())), 'device and oracle pick different patches'
            line['cpu_baseline'] = {'value': n * reps / dtc, 'unit': 'patches/s', 'cores': torch.get_num_threads(), 'kind': 'port',
                                    'sample': 'the whole pool, %d passes of batched forward + argsort in %.1f s' % (reps, dtc)}
        else:
            m = 1000

            class E(object):
                pars = {'patch_shape': in_shape[:2] + (1,)}
                nclass = 2
            osess = OracleSession(om)
            t0 = time.perf_counter()
            p = om.forward(xs[:m])['posteriors'][1].astype(np.float64)
            alpath.gen_A_matrices(E(), om, osess, xs[:m], p, 1e-5)
            dtc = time.perf_counter() - t0
            line['cpu_baseline'] = {'value': m / dtc, 'unit': 'patches/s', 'cores': torch.get_num_threads(), 'kind': 'port',
                                    'sample': '%d of the pool\'s patches, forward + per-sample gen_A_matrices (reference structure), %.1f s' % (m, dtc)}
    model.close()
    return line


def loop_config(args, sess, rank, ws):
    """configs[4]: the active-learning loop at the patch-tensor level (al_loop.run_rounds = the control flow of
    PW_AL.Experiment_MultiImg.run_method, PW_AL.py:690-898): 5 rounds over a pool of 200,000 synthetic 32^3 patches (sharded over the
    ranks), per round entropy filter to B = 4096 -> Fisher matrices -> SDP -> 100 draws -> fine-tune; value = patches scored by the
    filter per second of the whole loop, per-round stage seconds beside it."""
    import ctypes as C
    import torch
    from nnal_amd import al_loop, device, netspec, pool_shard
    from nnal_amd._lib import check
    n = 200000 if args.pool == 100000 else args.pool
    rounds = 5
    ld, sk = netspec.net_c()
    in_shape = (32, 32, 32, 1)
    pars = netspec.he_init(ld, in_shape, seed=15, skips=sk)
    model = device.DeviceModel(sess, ld, in_shape, sk, max_batch=args.batch)
    model.set_weights(pars)
    model.get_optimizer(1e-4, [], 'SGD')
    if not os.environ.get('ALQ_BENCH_NO_EVENTS'):
        model.lanes = 1       # the loop's launches are event-timed in place: one pipeline, so that a span is one launch
    a, b = pool_shard.shard_bounds(n, ws, rank)
    pool = sess.empty((b - a, 32 ** 3), torch.float32)
    check(sess.lib.alq_synth_patches(sess.ctx, 1005, a, b - a, 32 ** 3, C.c_void_p(pool.data_ptr())))
    lab_local = (pool[:, :512].sum(dim=1) > 0).cpu().numpy().astype(np.float64)
    labels = pool_shard.allgather_rows(n, np.arange(a, b), lab_local, sess).astype(np.int64)
    if ws > 1 and args.backend == 'nccl':
        try:
            pool_shard.attach_comm(sess)
        except Exception as e:
            print('[bench] rank %d: %s' % (rank, e), file=sys.stderr, flush=True)
    model.forward_device(pool, min(b - a, args.batch))           # warm-up pass
    warm = 'one forward pass'
    mw = min(b - a, 2 * args.batch)
    if args.warmup > 0 and mw * ws >= 1024 and pool_shard.shard_bounds(mw * ws, ws, rank) == (rank * mw, rank * mw + mw):
        # W = 1 untimed warm-up ROUND over the first two device passes of every shard: first-call costs of the stages behind the
        # filter (workspaces of the gradient kernels, optimiser state, the solver's first factorisation: ~0.4 s, a sixth of the
        # timed loop) belong to no round.  Weights and optimiser state are restored; the loop's own random stream is seeded per call.
        al_loop.run_rounds(model, sess, pool[:mw], 1, min(args.topB, mw // 2), 100, n_global=mw * ws, labels=labels[:mw * ws], finetune=dict(epochs=1, b=50))
        model.set_weights(pars)
        model.get_optimizer(1e-4, [], 'SGD')
        warm = 'one forward pass + one untimed round over the first %d patches of every shard (weights and optimiser state restored)' % mw
    sess.prof_reset()
    sess.prof_enable(0 if os.environ.get('ALQ_BENCH_NO_EVENTS') else args.prof_every)
    pool_shard.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = al_loop.run_rounds(model, sess, pool, rounds, args.topB, 100, n_global=n, labels=labels, finetune=dict(epochs=1, b=50))
    torch.cuda.synchronize()
    pool_shard.barrier()
    dt = pool_shard.max_over_ranks(time.perf_counter() - t0)
    sess.prof_enable(False)
    prof = sess.prof_read()
    scored = sum(rd['pool_left'] + len(rd['queries']) for rd in res)
    line = {'metric': 'patches/sec scored by the query loop (32^3, 2-class)', 'value': scored / dt, 'unit': 'patches/s', 'n_gpus': ws,
            'steps': rounds, 'warmup': 1, 'ms_per_step': 1e3 * dt / rounds, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'f32 (16-bit operand splits on the MFMA, fp32 accumulate; SDP and draws in fp64 on the host)', 'data': 'synthetic',
            'config': {'workload': 'configs[4]: active-learning loop, %d rounds over a pool of %d synthetic 32^3 patches, NET-C, per round entropy filter '
                                   '(B = %d) -> Fisher -> SDP -> 100 draws -> fine-tune (SGD, 1 epoch of batches of 50)' % (rounds, n, args.topB),
                       'pool_global': n, 'batch': model.max_batch, 'dist_world_size': ws, 'comm_world': int(getattr(sess, 'comm_world', 0) or 0),
                       'warmup_is': warm,
                       'rounds': [{'queries': int(len(rd['queries'])), 'pool_left': int(rd['pool_left']),
                                   'seconds': {k: float(v) for k, v in rd['seconds'].items()},
                                   'sdp': {k: (float(v) if isinstance(v, (int, float, np.floating)) else str(v)) for k, v in rd['sdp'].items()}}
                                  for rd in res]},
            'roofline': _conv_roofline(prof, 'the entropy filter (forward-only over the whole pool) is %.0f %% of the loop\'s wall time'
                                       % (100.0 * sum(rd['seconds']['filter'] for rd in res) / dt))}
    if not args.no_cpu_baseline and ws == 1 and rank == 0:
        from oracle.model import OracleModel
        torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 16)))
        om = OracleModel(ld, in_shape, pars, skips=sk)
        xs = pool[:256].cpu().numpy().reshape((-1,) + in_shape)
        om.forward(xs[:8])
        t0 = time.perf_counter()
        om.forward(xs)
        dtc = time.perf_counter() - t0
        line['cpu_baseline'] = {'value': len(xs) / dtc, 'unit': 'patches/s', 'cores': torch.get_num_threads(), 'kind': 'port',
                                'sample': 'batched oracle forward (the entropy filter, the loop\'s dominant stage) over %d pool patches, %.1f s; the '
                                          'Fisher stage of the reference structure runs at the configs[2] baseline rate' % (len(xs), dtc)}
    model.close()
    return line


def volume_config(args, sess, rank, ws):
    """--config 5: the query the reference actually times (PW_AL.Experiment_MultiImg.run_method, PW_AL.py:848-855):
    PW_NNAL.query_multimg(expr, model, sess, all_padded_imgs, pool_inds, labeled_inds, 'fi') (PW_NNAL.py:547-627) over padded
    VOLUMES - uncertainty filter over every pool voxel (get_patches gather -> channel normalisation -> NET-B forward), Fisher
    matrices of the B = 4096 most uncertain (gather with the slab statistics -> alq_fisher), SDP, k draws.  Synthetic subjects of
    the reference's shape: patch (25, 25, 1) x 2 modalities (run_on_subjects.py:18), NET-B = NN.create_PW1 (NN.py:1328-1336).
    value = pool voxels scored per second of the whole query; roofline = the gather kernel (HBM-bound: algorithmic bytes = the
    float64 voxels a patch reads + the float32 patch it writes) from a separate pass with HIP events around the gather calls."""
    import torch
    from nnal_amd import device, netspec, patch_utils, PW_NNAL
    S, m = 4, 2
    dims = (192, 192, 24)
    patch_shape = (25, 25, 1)
    rad = (12, 12, 0)
    rs = np.random.RandomState(1006)
    all_padded, pool_inds, stats = [], [], []
    for i in range(S):
        vols = [rs.randn(*dims) * (1. + .2 * j) + .3 * i for j in range(m)]                # float64, like nrrd.read + np.pad
        mask = (rs.rand(*dims) < .5).astype(np.int64)
        all_padded.append([np.pad(v, [(r, r) for r in rad], 'constant') for v in vols] + [mask])
        # grid of spacing 3 in plane, every slice (PW_AL.gen_multimg_inds, PW_AL.py:921-975): ~98k voxels per subject
        g = np.zeros(dims, bool)
        g[::3, ::3, :] = True
        pool_inds.append(np.nonzero(g.ravel())[0].astype(np.int64))
        stats.append([v for vol in vols for v in (float(vol.mean()), float(vol.std()))])
    n_pool = int(sum(len(p) for p in pool_inds))

    class Expr(object):
        pars = {'patch_shape': patch_shape, 'ntb': 8192, 'k': 100, 'B': args.topB, 'lambda_': 0., 'SDP_solver': 'CVXOPT'}
        nclass = 2
        train_stats = np.asarray(stats)
    expr = Expr()
    ld = netspec.net_b()
    in_shape = (25, 25, 2)
    pars = netspec.he_init(ld, in_shape, seed=16)
    model = device.DeviceModel(sess, ld, in_shape, (), max_batch=8192)
    model.set_weights(pars)
    np.random.seed(5)
    steps = max(1, args.steps)

    def step():
        return PW_NNAL.query_multimg(expr, model, sess, all_padded, pool_inds, [[] for _ in range(S)], 'fi')
    for _ in range(max(args.warmup, 1)):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        q = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # stage times of one more query (events / host clocks; not in `value`)
    t1 = time.perf_counter()
    dv = {}
    sel_inds, sel_posts = PW_NNAL.bin_uncertainty_filter_multimg(expr, model, sess, all_padded, pool_inds, expr.pars['B'], _vols=dv)
    torch.cuda.synchronize()
    t_filter = time.perf_counter() - t1
    # the gather alone: every pool voxel of every subject, chunks of 8192 as batch_eval cuts them, device events around the calls
    ev = []
    for i in range(S):
        for a in range(0, len(pool_inds[i]), 8192):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            t = dv[i].gather(pool_inds[i][a:a + 8192], patch_shape, expr.train_stats[i], quirk=1)
            e1.record()
            ev.append((e0, e1, int(t.shape[0])))
    torch.cuda.synchronize()
    g_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in ev)
    g_n = sum(c for _, _, c in ev)
    per_patch = 25 * 25 * m * (8 + 4)            # float64 voxels read + float32 patch written
    sess.prof_reset()
    sess.prof_enable(1)
    model.forward_device(t, int(t.shape[0]))
    torch.cuda.synchronize()
    sess.prof_enable(False)
    prof = sess.prof_read()
    ach = per_patch * g_n / (g_ms * 1e-3) / 1e9
    line = {'metric': 'pool voxels/sec scored by the volume-level fi query (25x25x1 x 2 modalities, 2-class)', 'value': n_pool * steps / dt, 'unit': 'patches/s',
            'n_gpus': 1, 'steps': steps, 'warmup': max(args.warmup, 1), 'ms_per_step': 1e3 * dt / steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64 gather + normalisation rounded once to f32; f32 network (16-bit operand splits on the MFMA, fp32 accumulate)',
            'data': 'synthetic',
            'config': {'workload': 'query_multimg(..., \'fi\') (PW_NNAL.py:547-627, timed by the reference at PW_AL.py:848-855) over %d synthetic subjects of %r '
                                   'voxels x %d modalities (float64, zero-padded by the patch radii), pool = in-plane grid of spacing 3 = %d voxels, patch %r, '
                                   'NET-B = NN.create_PW1 at [N, 25, 25, 2], B = %d, k = 100, SDP by NNAL_tools (cvxopt absent)' % (S, dims, m, n_pool, patch_shape, args.topB),
                       'pool_global': n_pool, 'batch': model.max_batch, 'queries_per_subject': [int(len(v)) for v in q],
                       'stage_seconds': {'uncertainty_filter_over_the_pool': t_filter, 'whole_query': dt / steps}},
            'roofline': {'bound': 'hbm', 'achieved': ach, 'peak': PEAK_HBM_TBPS * 1e3, 'unit': 'GB/s', 'frac': ach / (PEAK_HBM_TBPS * 1e3), 'traffic': None,
                         'kernel': 'gather_norm_kernel<double, float> (alq_gather_normalize = patch_utils.get_patches + the normalisation of PW_NN.batch_eval, '
                                   'patch_utils.py:1087-1173, PW_NN.py:503-506)',
                         'algorithmic_bytes_per_patch': per_patch, 'patches': g_n, 'launches': len(ev), 'avg_launch_ms': g_ms / max(len(ev), 1),
                         'note': 'torch events around DeviceVolumes.gather (index upload + one kernel) for every chunk of the pool; the windows of neighbouring '
                                 'grid voxels overlap, so most volume reads are L2 hits: the kernel is bound by its 5 KB of output per patch and by the '
                                 'per-chunk index upload, not by HBM reads',
                         'forward_pass_of_one_chunk_ms': {k: v['ms'] for k, v in prof.items() if v['ms'] > 0}}}
    if not args.no_cpu_baseline:
        from oracle import alpath
        from oracle.model import OracleModel, OracleSession
        torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 16)))
        om = OracleModel(ld, in_shape, pars)
        osess = OracleSession(om)
        nb = int(len(pool_inds[0]))          # ~12 s of the CPU port
        st0 = [[expr.train_stats[0, 2 * j], expr.train_stats[0, 2 * j + 1]] for j in range(m)]
        alpath.batch_eval(om, osess, all_padded[0][:-1], pool_inds[0][:64], patch_shape, 64, st0, 'posteriors')
        t0 = time.perf_counter()
        alpath.batch_eval(om, osess, all_padded[0][:-1], pool_inds[0][:nb], patch_shape, 256, st0, 'posteriors')
        dtc = time.perf_counter() - t0
        line['cpu_baseline'] = {'value': nb / dtc, 'unit': 'patches/s', 'cores': torch.get_num_threads(), 'kind': 'port',
                                'sample': 'the oracle\'s batch_eval (Python gather loop of get_patches + normalisation + batched forward, the filter '
                                          'stage that dominates the query) over %d pool voxels of subject 0, %.1f s' % (nb, dtc)}
    model.close()
    return line


NETB_BATCH = 2048      # same-box sweep over 8192 patches: 186 k patches/s at 256, 278 k at 512, 330 k at 1024, 350 k at 2048


def netb_rate(sess, n, x, accuracy=True):
    """"NET-B at [n,32,32,32] reported alongside" - the reference's literal patch net
    (NN.create_PW1: 42.05 M parameters, the 32 slices of a patch as channels) Fisher-scored on the first n pool
    patches, same outputs per patch; one warm-up pass, then one timed pass.  Not part of `value`."""
    import torch
    from nnal_amd import device, netspec
    ld = netspec.net_b()
    in_shape = (32, 32, 32)
    model = device.DeviceModel(sess, ld, in_shape, (), max_batch=NETB_BATCH)
    model.set_weights(netspec.he_init(ld, in_shape, seed=13))
    want = ('p1', 'H', 'g0', 'g1', 'A', 'trace', 'Asum')
    n = min(n, int(x.shape[0]))
    # warm-up over two passes when the timed call spans several: the second scoring pipeline (its own model, 42 M parameters packed on
    # the host) is created at the first call with more than one pass - outside the timed region
    model.fisher_device(x, min(n, 2 * NETB_BATCH), None, 1e-3, want=want)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.fisher_device(x, n, None, 1e-3, want=want)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # a third pass with HIP events around every launch (they cost a few per cent, so not the timed pass): the contraction
    # launches' own time for the roofline object
    lanes_timed = int(model.lanes)
    model.lanes = 1                       # one pipeline: an event span is one launch
    sess.prof_reset()
    sess.prof_enable(1)
    model.fisher_device(x, n, None, 1e-3, want=want)
    torch.cuda.synchronize()
    sess.prof_enable(False)
    prof = sess.prof_read()
    # the same accuracy statement as for NET-C, on the first 512 / 256 of these patches: every contraction launch of NET-B runs on
    # fp16 pairs (three products) - how far from the fp64 evaluation, next to the exact-fp32 engine on the same patches
    acc = None
    if accuracy:
        try:
            acc = accuracy_vs_exact_fp32(sess, model, x, min(n, 512), n64=min(n, 256))
        except Exception as e:      # an accuracy report must not take the throughput line with it
            acc = {'error': str(e)}
    model.close()
    bf = ('igemm_fwd', 'igemm_bwd', 'igemm3_fwd', 'igemm3_bwd', 'direct_conv')
    bf_ms, bf_fl = sum(prof[k]['ms'] for k in bf if k in prof), sum(prof[k]['flops'] for k in bf if k in prof)
    f16 = prof.get('igemm_f16x2', {'ms': 0.0, 'flops': 0.0, 'launches': 0})
    ms, fl = bf_ms + f16['ms'], bf_fl + f16['flops']
    nl = sum(prof[k]['launches'] for k in bf if k in prof) + f16['launches']
    peak_bf, peak_f16 = PEAK_BF16_MFMA_TFLOPS / SPLIT_PRODUCTS, PEAK_BF16_MFMA_TFLOPS / 3
    ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    peak = fl / (bf_fl / peak_bf + f16['flops'] / peak_f16) if fl > 0 else peak_bf
    traffic, traffic_note = None, None          # PMC passes are separate runs (PMC_ARGS="tools/gpu_netb.py 2048" tools/run_pmc.sh)
    tp = os.path.join(ROOT, 'profiles', 'netb_pmc_traffic.json')
    if os.path.exists(tp):
        try:
            tj = json.load(open(tp))
            traffic = tj['hbm_bytes_per_launch'] * NETB_BATCH / float(tj.get('batch', NETB_BATCH))
            traffic_note = 'PMC 2*FETCH_SIZE + WRITE_SIZE per contraction launch of a NET-B pass (profiles/netb_pmc_traffic.json)'
        except Exception:
            traffic = None
    return {'value': n / dt, 'unit': 'patches/s', 'net': 'NET-B = NN.create_PW1 (NN.py:1328-1336), input [N,32,32,32], 7 parameterised layers',
            'patches': n, 'batch': NETB_BATCH, 'scoring_pipelines': lanes_timed, 'flops_per_patch_executed': 190.9e6 + 151.5e6,
            'tflops_executed': (190.9e6 + 151.5e6) * n / dt / 1e12, 'accuracy': acc,
            'roofline': {'bound': 'mfma', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak if peak > 0 else 0.0,
                         'useful_frac': ach / peak_f16, 'traffic': traffic, 'traffic_note': traffic_note, 'launches': nl, 'avg_launch_ms': ms / max(nl, 1),
                         'kernel': 'the contraction launches of a NET-B pass (conv1 - conv3 forward and conv2 / conv4 backward-data on igemm3, conv4 forward '
                                   'as three 32-channel launches and conv3 backward-data on the two-slot engine, 2 wide fc forward + 2 wide fc backward '
                                   'on fcgemm; fp16 pairs wherever the launch knows a bound on its input - measured per-patch maxima pushed through the '
                                   'L1 norms forward, the cotangent bound backward - see alg_flops_per_patch), HIP events on every launch of a separate pass',
                         'peak_note': 'algorithmic fp32 flops; peak = flop-weighted harmonic mean of %.0f / 6 (bf16x3 launches, %.0f %% of the '
                                      'flops) and %.0f / 3 (fp16-pair launches)' % (PEAK_BF16_MFMA_TFLOPS, 100.0 * bf_fl / max(fl, 1.0), PEAK_BF16_MFMA_TFLOPS),
                         'alg_flops_per_patch': {'bf16x3': bf_fl / max(n, 1), 'f16x2': f16['flops'] / max(n, 1)},
                         'time_share_ms': {k: v['ms'] for k, v in prof.items()}}}


def accuracy_vs_exact_fp32(sess, model, x, n, n64=512):
    """Outside the timed region.  (1) One batch of the pool: the shipped engines against the exact-fp32 MFMA engine on the device
    (alq_debug_set(4, 1): fp32 fma chains, no operand split).  north_star's bar is 'scores within 1e-4': the scores are not
    continuous in the rounding noise (a ReLU input within rounding of zero switches a backward path), so what can be stated is how
    MANY patches differ by more than that between two fp32-level engines.  (2) `vs_fp64`: BOTH engines against the fp64 evaluation
    on the device (csrc/ref64.hip) on the first n64 patches: patches beyond 2e-6 / 1e-4, the largest difference, and how many
    fragile decisions (|pre-activation| <= eps x the layer's rms, pool near-ties) the fp64 evaluation has to invert to reproduce
    each engine's scores (0 / 1 / 2 / 3 / unexplained) - the evidence that the 16-bit splits are no further from the exact
    arithmetic than an fp32 engine is (tests/test_gpu_ref64.py asserts shipped <= 1.25 x exact-fp32 + 4)."""
    import torch
    from nnal_amd import ref64
    from nnal_amd._lib import check
    keys = ('p1', 'g0', 'g1')
    r = model.fisher_device(x, n, None, 1e-3, want=keys)
    a = {k: r[k].cpu().numpy().copy() for k in keys}
    check(sess.lib.alq_debug_set(4, 1))
    try:
        r = model.fisher_device(x, n, None, 1e-3, want=keys)
        b = {k: r[k].cpu().numpy().copy() for k in keys}
    finally:
        check(sess.lib.alq_debug_set(4, 0))
    torch.cuda.synchronize()
    d = np.maximum(np.abs(a['g0'] - b['g0']), np.abs(a['g1'] - b['g1'])).max(axis=1)
    out = {'patches': int(n), 'over_2e-6': int((d > 2e-6).sum()), 'over_1e-4': int((d > 1e-4).sum()), 'max_abs_dg': float(d.max()),
           'max_abs_dp': float(np.abs(a['p1'] - b['p1']).max()),
           'against': 'the exact-fp32 MFMA engine on the device (alq_debug_set(4, 1)), first batch of the pool; differences beyond 2e-6 are '
                      'ReLU / max-pool decisions that fp32 rounding puts on either side (each engine has its own set against fp64: vs_fp64)'}
    n64 = min(n64, n)
    if n64 > 0:
        t0 = time.perf_counter()
        r64 = ref64.Ref64(model, max_samples=128)
        rep, base, _ = r64.engine_report(x, np.arange(n64), {'shipped': (a['g0'][:n64], a['g1'][:n64]), 'exact_fp32': (b['g0'][:n64], b['g1'][:n64])},
                                         eps=ref64.DEFAULT_EPS)
        for k in ('shipped', 'exact_fp32'):
            rep[k]['max_abs_dp'] = float(np.abs((a if k == 'shipped' else b)['p1'][:n64] - base['p1']).max())
        out['vs_fp64'] = {'patches': int(n64), 'shipped': rep['shipped'], 'exact_fp32': rep['exact_fp32'], 'fragility': rep['_fragility'],
                          'seconds': time.perf_counter() - t0,
                          'how': 'fp64 forward + unit-cotangent backward + factored layer sums on the device (alq_ref64_scores); a patch whose '
                                 'scores are not within 2e-6 + 2e-5 relative of the plain fp64 value is re-evaluated with every set of <= 3 of '
                                 'its 10 most fragile decisions inverted (fragile: within eps x rms of the decision boundary); flips_needed = '
                                 'the fewest that reproduce the engine\'s scores'}
    return out


def cpu_baseline(xs, ld, sk, in_shape, pars, all_cores_too=False):
    """The oracle (CPU port with the reference's structure: batch 1, one backward per class per
    sample, full gradients materialised, NumPy shrink, PW_NNAL.py:757-814) on this box's host
    cores, on a bounded sample of the same patches."""
    import torch
    from oracle import alpath
    from oracle.model import OracleModel, OracleSession
    # the box's CPU share, not the host's core count (a cgroup-limited container reports all
    # host cores in os.cpu_count(); oversubscribing them stalls OpenMP)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    torch.set_num_threads(cores)
    om = OracleModel(ld, in_shape, pars, skips=sk)
    osess = OracleSession(om)
    xs = xs.reshape((-1,) + in_shape)

    class E(object):
        pars = {'patch_shape': in_shape[:3]}
        nclass = 2
    p = om.forward(xs[:2])['posteriors'][1]
    alpath.gen_A_matrices(E(), om, osess, xs[:2], p.astype(np.float64), 1e-3)    # warm-up
    t0 = time.perf_counter()
    p = om.forward(xs)['posteriors'][1].astype(np.float64)
    alpath.gen_A_matrices(E(), om, osess, xs, p, 1e-3)
    dt = time.perf_counter() - t0
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = None
    # ... and the same port with every core the process may use (os.cpu_count() threads), on a shorter sample: the
    # reference's structure is batch 1, so beyond ~16 threads a sample gains little (the per-op work is one patch)
    all_cores = None
    nall = max(1, min(affinity or (os.cpu_count() or 1), os.cpu_count() or 1))
    if nall > cores and all_cores_too:
        used = torch.get_num_threads()
        torch.set_num_threads(nall)
        # ONE patch, no warm-up: at batch 1 with hundreds of threads every op is oversubscribed (measured on the 256-thread GPU box:
        # ~34 s per patch against 0.12 s at 16 threads) - the number is reported because BASELINE.md names os.cpu_count() threads
        t1 = time.perf_counter()
        m = 1
        p2 = om.forward(xs[:1])['posteriors'][1].astype(np.float64)
        alpath.gen_A_matrices(E(), om, osess, xs[:1], p2, 1e-3)
        dt2 = time.perf_counter() - t1
        all_cores = {'value': m / dt2, 'unit': 'patches/s', 'cores': torch.get_num_threads(), 'sample': '%d patches, %.1f s' % (m, dt2)}
        torch.set_num_threads(used)
    return {'value': len(xs) / dt, 'unit': 'patches/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'host_cores': os.cpu_count(), 'affinity_cores': affinity, 'all_cores': all_cores,
            'sample': '%d of the pool\'s patches (NET-C 32^3), forward + per-sample gen_A_matrices, %.1f s; %d threads '
                      '(the box reports %s cores, %s in this process\'s affinity mask).  BASELINE.md 3 names os.cpu_count() threads: '
                      'the reference structure is batch 1, every op is then oversubscribed (0.03 patches/s at 256 threads, rounds 4-5; '
                      '--cpu-all-cores re-measures it into `all_cores`), so the baseline stated here is the FASTER 16-thread one'
                      % (len(xs), dt, torch.get_num_threads(), os.cpu_count(), affinity)}


if __name__ == '__main__':
    main()
