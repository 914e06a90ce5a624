"""bench.py -- throughput of the MI355X hot path on synthetic 10 s @ 16 kHz x 32 batches.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload asr_train_step|frontend|tts_train_step|asr_decode]

One process per GPU.  N > 1: either launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE from the environment), or -- with no
launcher around it -- this script starts its N ranks itself as child processes (`self_launch`).  A step is one pass of the hot path over
one per-GPU batch of 32 synthetic utterances; inputs are resident in HBM before the timed region; every timed step of the train-step
workloads replays the same hipGraphs.  The path shards by utterance (weak scaling): the only exchanges are the gradient all-reduce, the
feature-normalisation statistics and the BatchNorm statistics (SyncBatchNorm, as the reference's DDP set-up).  Rank 0 prints ONE JSON line
(see the repo prompt's contract) with two extra objects:
  roofline     -- the dominant kernel against the bound that applies to it: for the train steps the GEMM kernel with the largest summed
                  time vs the 2.5 PFLOP/s dense bf16 MFMA peak (flops / HIP-event time of ONE extra eager, instrumented step that runs right
                  AFTER the timed region); for the frontend its algorithmic bytes / HIP-event launch time vs the 8 TB/s HBM peak
  cpu_baseline -- the CPU oracle (a torch-CPU port of the reference op sequence) timed on this host's cores on a
                  bounded sample of the same workload (rank 0, N = 1 only)
"""
import argparse
import gc
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling

ASR_FRONTEND = dict(sr=16000, preemphasis=0.97, hop_length=0.010, win_length=0.025, n_mels=80)
SPECAUG = dict(freq_mask_width=15, freq_mask_num=4, time_mask_width=0.05, time_mask_num=5)
BATCH, SAMPLES = 32, 160000


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (100 timed steps by default, under a second of replayed graphs; the eager, event-instrumented step of the roofline runs AFTER them)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default="asr_train_step", choices=["asr_train_step", "frontend", "tts_train_step", "asr_decode"],
                    help="asr_train_step = the headline metric (BASELINE configs[2]); frontend = configs[1]; tts_train_step = configs[4] "
                         "(FastSpeech2 train step, 32 x 7.4 s @ 22.05 kHz: a secondary line with its own metric name); asr_decode = SURVEY "
                         "row f3 (encoder + beam search, beam 16, CTC weight 0.2: a secondary line with its own metric name)")
    ap.add_argument("--prenet-dims", default="64,64",
                    help="conv channels of the Conv2d prenet in front of the encoder (SURVEY row f2; HIP path, FLOPs outside the named-path count). "
                         "64,64 = the transformer-wide recipes (and the module default); 512,512 = the transformer-large recipes (conv_dims = d_model)")
    ap.add_argument("--launch-check", action="store_true",
                    help="only bring the N ranks up (RCCL process group, or gloo where there is no GPU), all-reduce one number, print "
                         '{"launch_check": "ok", ...} and exit: what tests/test_bench_launch.py runs on CPU')
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def synth_batch(seed: int, device):
    g = torch.Generator().manual_seed(seed)
    wave = (torch.randn(BATCH, SAMPLES, 1, generator=g) * 0.1)
    lens = torch.full((BATCH,), SAMPLES, dtype=torch.int64)
    return wave.to(device), lens.to(device), lens


class FrontendWorkload:
    """BASELINE.json configs[1]: waveform -> log-mel (fused HIP kernel) -> SpecAugment (warp + masks)."""

    name = "frontend+specaug 32x10s@16kHz (BASELINE configs[1])"
    dtype = "f32"
    # algorithmic HBM bytes of the dominant kernel (the fused frontend) per launch: read the waveform once, write the
    # log-mel once: 32*160000*4 + 32*1001*80*4
    alg_bytes = BATCH * SAMPLES * 4 + BATCH * 1001 * 80 * 4
    dominant = "frontend_pfa400_kernel"

    def __init__(self, device, seed):
        from speechain_amd.module.augment.specaug import SpecAugment
        from speechain_amd.module.frontend.speech2mel import Speech2MelSpec

        self.fe = Speech2MelSpec(**ASR_FRONTEND).to(device)
        self.sa = SpecAugment(input_size=80, **SPECAUG).to(device)
        self.sa.train()
        self.wave, self.lens, self.lens_host = synth_batch(seed, device)
        # The lengths are known on the host (a data loader made them): SpecAugment gets the frame counts as a CPU tensor and draws its
        # indices on the host generator -- no device-to-host read, one pinned copy of the index tensors (module/augment/specaug.py).
        # BENCH_DEVICE_DRAWS=1: device-resident lengths and draws on the device generator (three host round trips per step).
        self.host_lens = os.environ.get("BENCH_DEVICE_DRAWS", "0") != "1"
        self.sa.host_draws = self.host_lens
        self.flen_host = self.fe.speech2linear.frame_count(self.lens_host)
        self.ev = []

    def step(self, timed: bool):
        from speechain_amd._lib import lib

        # HIP events are recorded on the launch stream tightly around the dominant kernel's launch
        lib.timing = {"sc_frontend_logmel_fwd": self.ev} if timed else None
        feat, flen = self.fe(self.wave, self.lens)
        lib.timing = None
        out, _ = self.sa(feat, self.flen_host if self.host_lens else flen)
        return out

    def dominant_ms(self):
        return sum(a.elapsed_time(b) for a, b in self.ev) / max(1, len(self.ev))


NAMED_PATH_FLOPS_PER_STEP = 2.54e12
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 matrix-core peak of MI355X (MI355X_MICROARCH.md)
ASR_MODEL = dict(d_model=512, num_heads=8, enc_layers=12, dec_layers=6, fdfwd_dim=2048, vocab_size=5000, dropout=0.1,
                 activation="GELU", label_smoothing=0.2, ctc_weight=0.3)
TEXT_LEN = 64


def synth_text(seed: int, device, vocab: int = 5000):
    g = torch.Generator().manual_seed(1000 + seed)
    text = torch.randint(1, vocab - 2, (BATCH, TEXT_LEN), generator=g)
    text[:, 0] = vocab - 1
    text[:, -1] = vocab - 1
    return text.to(device), torch.full((BATCH,), TEXT_LEN, dtype=torch.int64, device=device)


PRENET_NOTE = ("Conv2dPrenet (k3 s2 x2, no bias + BatchNorm2d + LeakyReLU + Linear; SURVEY.md row f2) runs on the HIP path too "
               "(csrc/prenet.hip + the library's GEMMs, bf16); its FLOPs are NOT counted in roofline.step_named_path. "
               "Both channel settings occur in the reference's ASR recipes for this d512/h8/ffn2048 transformer: conv_dims [64,64] "
               "in the transformer-wide_ctc_perturb recipes (recipes/asr/librispeech/train-960/exp_cfg/960-bpe5k_transformer-wide_ctc_"
               "perturb.yaml:205-213; also the module default), conv_dims = d_model = [512,512] in the transformer-large recipes "
               "(.../train-clean-100/exp_cfg/100-bpe5k_transformer-large_lr2e-3.yaml:197-206), whose convolutions cost more FLOPs than "
               "the 12-layer encoder. SURVEY.md section 8d: [64,64] primary (keeps the step dominated by the named path), [512,512] "
               "secondary: run --prenet-dims 512,512 (measured: profiles/r01_train_step_bench_prenet512.json)")


def prenet_conf(dims):
    """The transformer recipes' enc_prenet block (100-bpe5k_transformer-large_lr2e-3.yaml:196-206) with the given channels."""
    return dict(conv_dims=list(dims), conv_kernel=3, conv_stride=2, conv_batchnorm=True, conv_activation="LeakyReLU", lnr_dims=512)


class no_gc:
    """No cyclic garbage collection inside the block: a pass that frees graphs or device tensors while a stream is capturing calls into
    the runtime in the middle of the capture, which aborts the process (seen in a test run)."""

    def __enter__(self):
        self.was = gc.isenabled()
        gc.disable()

    def __exit__(self, *exc):
        if self.was:
            gc.enable()
        return False


def syncbn_group():
    """A communicator of their own for the BatchNorm statistics.  Their all-reduces are captured into the step's hipGraphs and run on the
    main stream in graph order, identically on every rank; on the DEFAULT group they would share a communicator with the eager all-gather of
    the next batch's normalisation statistics (head stream), and two collectives of one communicator that the ranks may start in different
    orders are a deadlock.  (None -- the default group -- if the group cannot be made.)"""
    import torch.distributed as dist
    try:
        return dist.new_group(backend="nccl")
    except Exception as e:
        print(f"bench: dist.new_group failed ({e}); BatchNorm statistics stay on the default process group", file=sys.stderr)
        return None


class ASRTrainStepWorkload:
    """BASELINE.json configs[2] driven from waveforms: frontend -> global mean/std normalisation -> SpecAugment -> Conv2dPrenet (row f2; channels
    from --prenet-dims) -> 12-layer encoder -> 6-layer decoder -> CTC + CE heads -> backward -> grad clip -> Adam (Noam), bf16 compute."""

    dtype = "bf16"

    def __init__(self, device, seed, world, prenet_dims=(64, 64), distributed=False):
        self.prenet_dims = tuple(prenet_dims)
        self.name = ("ASR transformer-base train step (12enc/6dec d512 h8 ffn2048 V5000, bf16) from 32x10s@16kHz waveforms: frontend + "
                     f"global mean/std normalisation + specaug + conv2d-prenet{list(self.prenet_dims)} + encoder + decoder + "
                     "0.7CE/0.3CTC + backward + clip + Adam "
                     "(BASELINE configs[2])")
        from speechain_amd import functional as Fn
        from speechain_amd.ddp import GradBucketReducer, attention_fusion_groups
        from speechain_amd.harness import ASRTransformer
        from speechain_amd.module.augment.specaug import SpecAugment
        from speechain_amd.module.frontend.speech2mel import Speech2MelSpec
        from speechain_amd.module.norm.feat_norm import FeatureNormalization
        from speechain_amd.module.prenet.conv2d import Conv2dPrenet

        Fn.set_compute_dtype(torch.bfloat16)
        Fn.set_attention_probs(False)  # the train step never reads the attention maps: fused attention, no score tensors in HBM
        torch.manual_seed(0)  # identical initial weights on every rank
        self.fe = Speech2MelSpec(**ASR_FRONTEND).to(device)
        # the recipes' `normalize: True` (global mean / std); under DDP its statistics are gathered over the ranks every step, as the
        # reference does (module/encoder/asr.py:55-57 passes `distributed`; feat_norm.py:236-265) -- they are buffers, so nothing else
        # would keep the ranks' normalisation identical
        self.norm = FeatureNormalization(input_size=80, distributed=distributed).to(device).train()
        self.sa = SpecAugment(input_size=80, **SPECAUG).to(device).train()
        self.prenet = Conv2dPrenet(input_size=80, **prenet_conf(self.prenet_dims)).to(device).train()
        if distributed and os.environ.get("BENCH_SYNCBN", "1") == "1":
            # speechain/runner.py:1856-1858 (`enable_syncbatchnorm`, default True): every BatchNorm of a DDP model normalises with the
            # statistics of the batches of ALL ranks -- the HIP prenet all-reduces its C-sized sums between its reduce and finish kernels
            self.prenet = torch.nn.SyncBatchNorm.convert_sync_batchnorm(self.prenet, process_group=syncbn_group())
        self.model = ASRTransformer(**ASR_MODEL).to(device).train()
        for m in (self.prenet, self.model):
            for p in m.parameters():
                if p.dim() > 1:
                    torch.nn.init.xavier_normal_(p)
        self.params = [p for m in (self.prenet, self.model) for p in m.parameters()]
        # The gradient buckets travel on a communicator of their own: torch keeps ONE stream per (device, process group) for its
        # collectives, so on the default group the small all-gather of the next batch's normalisation statistics (head stream, followed by
        # SpecAugment's host-side reads) would queue behind this step's bucket all-reduces and tie the host to the end of backward.
        grad_pg = None
        if distributed and os.environ.get("BENCH_GRAD_PG", "1") == "1":
            import torch.distributed as dist
            try:
                grad_pg = dist.new_group(backend="nccl")
            except Exception as e:  # (a second communicator is an optimisation, never a reason to lose the run)
                print(f"bench: dist.new_group failed ({e}); gradient buckets stay on the default process group", file=sys.stderr)
                grad_pg = None
        self.reducer = GradBucketReducer(self.params, bucket_mb=32.0, process_group=grad_pg, adjacent=attention_fusion_groups(self.model))
        self.reducer.broadcast_parameters()
        self.lr = torch.tensor([2e-3], device=device)
        self.noam_warmup = 25000  # (tests/test_bench_path_gpu.py shortens it so that a handful of steps moves the weights)
        # clip at 5.0 + Adam(0.9, 0.98, 1e-9) over the flat buckets in two launches (csrc/optim.hip); BENCH_TORCH_ADAM=1: stock torch ops
        self.fused_opt = os.environ.get("BENCH_TORCH_ADAM", "0") != "1"
        if self.fused_opt:
            from speechain_amd.optim import FusedAdam
            self.opt = FusedAdam(self.reducer, self.lr, betas=(0.9, 0.98), eps=1e-9, max_grad_norm=5.0)
        else:
            self.opt = torch.optim.Adam(self.params, lr=self.lr, betas=(0.9, 0.98), eps=1e-9, fused=True, capturable=True)
        torch.manual_seed(1234 + seed)
        self.wave, self.lens, lens_host = synth_batch(seed, device)
        # (as in FrontendWorkload: the data loader knows the lengths on the host, so SpecAugment draws its indices on the host generator from the
        # host-side frame counts -- no device-to-host reads and no one-workgroup randint launches on the prefetch stream; BENCH_DEVICE_DRAWS=1: as before)
        self.host_lens = os.environ.get("BENCH_DEVICE_DRAWS", "0") != "1"
        self.sa.host_draws = self.host_lens
        self.flen_host = self.fe.speech2linear.frame_count(lens_host)
        self.text, self.text_len = synth_text(seed, device)
        self.feat = torch.zeros(BATCH, 1001, 80, device=device)      # static inputs of the captured region
        self.flen = torch.zeros(BATCH, dtype=torch.int64, device=device)
        self.nstep = 0
        self.ev, self.flops = [], 0.0
        self.launch_flops, self.launch_kernels = [], []  # per timed launch, same order as self.ev
        self.world = world
        self.g_body = self.g_tail = None
        self.loss = None
        self.use_graph = os.environ.get("BENCH_NO_GRAPH", "0") != "1"
        # more than one rank: backward is captured in pieces and each piece's finished buckets are all-reduced under the next one
        self.segmented = self.reducer.multi and os.environ.get("BENCH_NO_SEGMENTS", "0") != "1"
        self.g_pieces, self.piece_buckets = [], []
        # The head of a step (frontend -> normalisation -> SpecAugment; eager, its index draws read lengths on the host) depends on the
        # data only, so the head of batch i+1 runs on a second stream while the replayed graphs of batch i occupy the main one: the
        # host-bound 0.3 ms no longer leave the GPU idle.  Every timed step still runs one head and one body (the prefetch of the batch
        # after the last timed step is inside the timed region; the very first head ran during warm-up).  BENCH_NO_PREFETCH=1: in line.
        self.prefetch = os.environ.get("BENCH_NO_PREFETCH", "0") != "1"
        self.head_stream = torch.cuda.Stream(device=device) if self.prefetch else None
        self.next_head = None  # (feat, flen, event) of the batch the next step trains on

    def _head(self):
        with torch.no_grad():  # frontend + SpecAugment stay eager: SpecAugment's index draws read the lengths on the host
            feat, flen = self.fe(self.wave, self.lens)
            feat, flen = self.norm(feat, flen, epoch=1)  # module/encoder/asr.py:111-115: between the frontend and SpecAugment
            if self.host_lens:
                out, _ = self.sa(feat, self.flen_host)
                return out, flen
            return self.sa(feat, flen)

    def _load_batch(self):
        """Features of the batch this step trains on -> the static inputs of the captured region."""
        main = torch.cuda.current_stream()
        if not self.prefetch:
            feat, flen = self._head()
        else:
            if self.next_head is None:  # first step: nothing was prefetched
                with torch.cuda.stream(self.head_stream):
                    self.head_stream.wait_stream(main)
                    feat, flen = self._head()
                    ev = torch.cuda.Event()
                    ev.record(self.head_stream)
                self.next_head = (feat, flen, ev)
            feat, flen, ev = self.next_head
            main.wait_event(ev)
        with torch.no_grad():
            self.feat.copy_(feat)
            self.flen.copy_(flen)
        if self.prefetch:  # the head stream may reuse the memory of (feat, flen) only after these copies have run
            self.copied = torch.cuda.Event()
            self.copied.record(main)

    def _prefetch_next(self):
        """Called right after this step's device work has been launched: the next batch's head, on the head stream."""
        if not self.prefetch:
            return
        with torch.cuda.stream(self.head_stream):
            self.head_stream.wait_event(self.copied)
            feat, flen = self._head()
            ev = torch.cuda.Event()
            ev.record(self.head_stream)
        self.next_head = (feat, flen, ev)

    # ---- the two device-side halves of a step; eager or replayed from hipGraphs --------------------------------------
    def _body(self):
        """zero grads -> prenet -> encoder/decoder/heads/losses -> backward (parameter grads land in the flat buckets)."""
        from speechain_amd import functional as Fn

        Fn.DropoutState.advance()
        Fn.weights.refresh()  # one launch re-casts every bf16 weight copy (plain, fused q|k|v, transposed)
        self.reducer.zero_grad()
        x, xlen = self.prenet(self.feat, self.flen)
        out = self.model(x, xlen, self.text, self.text_len)
        out["loss"].backward()
        # gradients that stock torch ops produced (token embedding, the permuted prenet weights) are copied into their bucket slots HERE, inside the
        # region that is captured: the clip below works on the flat buckets and the all-reduce ships them, so after a replay the
        # slots must hold this step's values (tests/test_bench_step_gpu.py)
        self.reducer.settle()
        return out["loss"].detach()

    # ---- multi-rank: the same work in pieces, so that finished gradient buckets travel while the rest of backward runs ----------
    def _piece_params(self):
        """Parameters whose gradients each backward piece produces; the cut tensors are the encoder output, two encoder hidden states and
        the encoder input (the prenet's output)."""
        m = self.model
        enc = m.encoder
        late = [m.decoder, m.embed, m.postnet, m.ctc_layer]
        groups = [[p for mod in late for p in mod.parameters()],
                  [p for l in enc.trfm_layers[8:] for p in l.parameters()] + list(enc.layernorm.parameters()),
                  [p for l in enc.trfm_layers[4:8] for p in l.parameters()],
                  [p for l in enc.trfm_layers[:4] for p in l.parameters()] + list(enc.posenc.parameters()),
                  list(self.prenet.parameters())]
        assert sum(len(g) for g in groups) == len(self.params)
        return [[p for p in g if p.requires_grad] for g in groups]

    def _pieces(self):
        """The device work of `_body` as five callables sharing state: forward + backward down to the encoder output, then backward
        through encoder layers 11..8, 7..4, 3..0 and the prenet.  Each ends where `torch.autograd.backward(..., inputs=...)` is told to
        stop; the gradient that crosses a cut is accumulated in the cut tensor's .grad and handed to the next piece."""
        from speechain_amd import functional as Fn

        st = {}
        P = self._piece_params()

        def first():
            Fn.DropoutState.advance()
            Fn.weights.refresh()
            self.reducer.zero_grad()
            x, xlen = self.prenet(self.feat, self.flen)
            x = x.view_as(x)  # cut tensors must come from a metadata-only node (see TransformerEncoder.forward)
            out = self.model(x, xlen, self.text, self.text_len)
            hid = out["enc_hidden"]
            st["cuts"] = [out["enc"], hid[7], hid[3], x]
            st["loss"] = out["loss"].detach()
            torch.autograd.backward([out["loss"]], inputs=P[0] + [st["cuts"][0]])

        def middle(k):
            def run():
                cut = st["cuts"][k - 1]
                nxt = [st["cuts"][k]] if k < len(st["cuts"]) else []
                torch.autograd.backward([cut], grad_tensors=[cut.grad], inputs=P[k] + nxt)
                if not nxt:
                    self.reducer.settle()
            return run

        return [first] + [middle(k) for k in range(1, 5)], st

    def _tail(self):
        """global-norm clip at 5.0 (no host sync) -> Adam."""
        if self.fused_opt:
            self.opt.step()
            return
        flats = [b["flat"] for b in self.reducer.buckets]
        norm = torch.sqrt(sum((f * f).sum() for f in flats))
        coef = torch.clamp(5.0 / (norm + 1e-6), max=1.0)
        for f in flats:
            f.mul_(coef)
        self.opt.step()

    def _capture(self):
        from speechain_amd import functional as Fn

        # Every collective of the eager warm-up steps is complete before capture starts, and capture polices only THIS thread:
        # RCCL's watchdog thread keeps polling completion events (hipEventQuery), which the default global capture mode
        # rejects from any thread while a capture is open ("operation not permitted when stream is capturing").
        self.reducer.wait()
        torch.cuda.synchronize()
        if self.segmented:
            # one hipGraph per piece, all in one memory pool (activations saved by the first piece are consumed by the later ones); after
            # each capture the reducer tells which buckets that piece completed -- at replay time their all-reduces are issued right
            # after that piece, on the communication stream, while the next piece runs
            pieces, st = self._pieces()
            self.g_pieces, self.piece_buckets, taken, pool = [], [], set(), None
            for run in pieces:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                    run()
                pool = g.pool()
                self.g_pieces.append(g)
                self.piece_buckets.append(self.reducer.completed_buckets(taken))
            self.piece_buckets[-1] += [i for i in range(len(self.reducer.buckets)) if i not in taken]  # (buckets holding unused parameters)
            self.loss = st["loss"]
            self.g_body = self.g_pieces[0]
        else:
            self.g_body = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_body, capture_error_mode="thread_local"):
                self.loss = self._body()
        self.g_tail = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_tail, capture_error_mode="thread_local"):
            self._tail()
        torch.cuda.synchronize()

    def step(self, timed: bool, instrument: bool = False):
        from speechain_amd import functional as Fn
        from speechain_amd._lib import lib

        self.nstep += 1
        # Noam schedule, warmup 25000 (optim_sche/noam.py:42-59): a scalar fill outside the graphs
        self.lr.fill_(2e-3 * min(self.nstep ** -0.5, self.nstep * self.noam_warmup ** -1.5) * self.noam_warmup ** 0.5)
        self._load_batch()
        graph = self.use_graph and not instrument and self.nstep > 3
        if graph and self.g_body is None:
            try:
                with no_gc():
                    self._capture()
            except Exception as e:  # (e.g. a runtime that refuses a collective inside a capture: keep the run, eagerly, and say so)
                print(f"bench: capturing the step into hipGraphs failed ({type(e).__name__}: {e}); continuing with eager steps", file=sys.stderr)
                self.use_graph, graph = False, False
                self.g_body = self.g_tail = None
                self.g_pieces, self.piece_buckets = [], []
                torch.cuda.synchronize()
        if graph and self.segmented:
            for g, ready in zip(self.g_pieces, self.piece_buckets):
                g.replay()
                for bi in ready:
                    self.reducer.launch_bucket(bi)
            self.reducer.wait()  # the optimizer's stream waits for the collectives; nothing is launched twice
            self.g_tail.replay()
            self._prefetch_next()
            return self.loss
        if graph:
            self.g_body.replay()
            if self.reducer.multi:
                self.reducer.reduce_all_now()
            self.g_tail.replay()
            self._prefetch_next()
            return self.loss
        if instrument:  # HIP events around every GEMM launch of this (eager) step
            lib.timing = {"sc_gemm": self.ev, "sc_gemm_splitk": self.ev, "sc_wgrad_grouped": self.ev}
            lib.flop_log = []
            lib.kernel_log = []
            lib.shape_log = [] if os.environ.get("BENCH_DUMP_GEMMS") else None
        loss = self._body()
        self.reducer.wait()
        self._tail()
        if instrument:
            self.flops += sum(lib.flop_log)
            self.launch_flops += lib.flop_log
            self.launch_kernels += lib.kernel_log
            if lib.shape_log is not None:  # BENCH_DUMP_GEMMS=1: per-shape table of the instrumented step on stderr
                torch.cuda.synchronize()
                tab = {}
                for (a, b2), shp, kid in zip(self.ev[-len(lib.shape_log):], lib.shape_log, lib.kernel_log):
                    e = tab.setdefault((kid,) + shp, [0, 0.0])
                    e[0] += 1
                    e[1] += a.elapsed_time(b2)
                for key, (c, t) in sorted(tab.items(), key=lambda kv: -kv[1][1]):
                    kid, M_, N_, K_, nb_ = key
                    print(f"GEMM kernel={kid} M={M_} N={N_} K={K_} batch={nb_}: {c} launches, {t * 1e3 / c:.1f} us each, {t:.3f} ms, "
                          f"{2.0 * M_ * N_ * K_ * nb_ * c / (t * 1e-3) / 1e12:.0f} TFLOP/s", file=sys.stderr)
                lib.shape_log = None
            lib.timing = None
            lib.flop_log = None
            lib.kernel_log = None
        self._prefetch_next()
        return loss

    def roofline(self):
        """The dominant GEMM kernel of the step (largest summed launch time) against the dense bf16 MFMA peak, from HIP events
        around every sc_gemm / sc_gemm_splitk call of the one eager step; `all_gemm_launches` is the same over all of them."""
        from speechain_amd._lib import lib

        times = [a.elapsed_time(b) for a, b in self.ev]
        ms, n = sum(times), len(times)
        agg = self.flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        per = {}
        for t, f, k in zip(times, self.launch_flops, self.launch_kernels):
            e = per.setdefault(k, [0.0, 0.0, 0])
            e[0] += t
            e[1] += f
            e[2] += 1
        by_kernel = {lib.load().sc_gemm_kernel_name(k).decode(): {"launches": c, "ms": round(t, 4), "avg_launch_ms": round(t / c, 5),
                                                                   "tflops": round(f / (t * 1e-3) / 1e12, 2) if t > 0 else 0.0}
                     for k, (t, f, c) in sorted(per.items(), key=lambda kv: -kv[1][0])}
        # the headline kernel must be ONE kernel (its average is checked against the rocprofv3 row of that name): the split-K
        # entry points launch two (product + splitk_reduce_kernel) inside one timed interval, so they are listed but not eligible
        single = {k: v for k, v in per.items() if k not in (8, 9, 10, 13, 14)}  # (a grouped wgrad call may be 2-3 launches)
        dom_id, (dt, df, dc) = max(single.items(), key=lambda kv: kv[1][0]) if single else (0, (0.0, 0.0, 0))
        achieved = df / (dt * 1e-3) / 1e12 if dt > 0 else 0.0
        return {"bound": "mfma", "kernel": lib.load().sc_gemm_kernel_name(dom_id).decode(), "achieved": round(achieved, 2),
                "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 5),
                # (the committed PMC passes are of the ASR step: its per-launch averages describe that step's mix of shapes only)
                "traffic": pmc_traffic(lib.load().sc_gemm_kernel_name(dom_id).decode()) if isinstance(self, ASRTrainStepWorkload) else None,
                "mfma_busy_pmc": pmc_mfma_busy(lib.load().sc_gemm_kernel_name(dom_id).decode()) if isinstance(self, ASRTrainStepWorkload) else None,
                "launches_timed": dc, "avg_launch_ms": round(dt / max(dc, 1), 5), "kernel_flops_timed": df,
                "all_gemm_launches": {"achieved": round(agg, 2), "frac": round(agg / MFMA_BF16_PEAK_TFLOPS, 5), "launches_timed": n,
                                      "avg_launch_ms": round(ms / max(n, 1), 5), "gemm_flops_timed": self.flops},
                "by_kernel": by_kernel,
                "step_named_path": None,  # filled by main(): SURVEY 8d's 2.54 TFLOP of the named path / the measured step time
                "note": "per-kernel times: ONE extra step right after the timed region runs eagerly with HIP events around every "
                        "sc_gemm/sc_gemm_splitk launch (the timed steps all replay hipGraphs; `value`/`ms_per_step` do not contain it); `kernel` is the single GEMM kernel with the largest summed time in that step, achieved = its "
                        "2*M*N*K*batch flops / its event time; the split-K entries of by_kernel time two launches (product + "
                        "splitk_reduce_kernel) in one interval and are therefore not eligible as `kernel`; traffic: "
                        "HBM bytes per launch of `kernel` from the committed PMC passes (profiles/r03_train_step_pmc_traffic.json, or the newest earlier one: "
                        "FETCH_SIZE x 2 + WRITE_SIZE over eager steps, averaged over the mix of shapes the kernel serves); mfma_busy_pmc: its matrix-core busy "
                        "fraction from the committed SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE pass (profiles/r03_train_step_pmc_mfma.json, or the newest earlier one)"}


# ---- BASELINE configs[4]: FastSpeech2 train step -----------------------------------------------------
TTS_SAMPLES, TTS_FRAMES, TTS_TOKENS, TTS_VOCAB = 163584, 640, 128, 75
TTS_DROPOUT = (0.2, 0.5)  # trmf_dropout, conv_dropout of the recipe (tests set (0, 0) to compare replayed and eager steps)


def tts_recipe_conf(dropout_trf=0.2, dropout_conv=0.5):
    """module_conf of recipes/tts/ljspeech/exp_cfg/22.05khz_mfa_fastspeech2.yaml:150-249 (d 384, 2 heads, 4 + 4 layers, conv feed-forward
    1536 k3, predictors 256 channels, postnet 256 x 4 + back to 80, trmf_dropout 0.2, conv_dropout 0.5)."""
    trf = dict(posenc_dropout=dropout_trf, posenc_scale=True, emb_scale=False, emb_layernorm=True, d_model=384, num_heads=2, num_layers=4,
               att_dropout=dropout_trf, fdfwd_dim=1536, fdfwd_type="conv", fdfwd_args=dict(kernel_size=3), fdfwd_dropout=dropout_trf,
               res_dropout=dropout_trf, layernorm_first=True)
    vp = "prenet.var_pred.Conv1dVarPredictor"
    return dict(
        enc_emb=dict(type="prenet.embed.EmbedPrenet", conf=dict(embedding_dim=384)),
        encoder=dict(type="transformer.encoder.TransformerEncoder", conf=dict(trf)),
        duration_predictor=dict(type=vp, conf=dict(conv_dims=[256, -1], conv_kernel=3, conv_dropout=dropout_conv)),
        pitch_predictor=dict(type=vp, conf=dict(conv_dims=[256, -1, -1, -1, -1], conv_kernel=5, conv_dropout=dropout_conv, conv_emb_kernel=1,
                                                conv_emb_dropout=0.0)),
        energy_predictor=dict(type=vp, conf=dict(conv_dims=[256, -1], conv_kernel=3, conv_dropout=dropout_conv, conv_emb_kernel=1,
                                                 conv_emb_dropout=0.0)),
        feat_frontend=dict(type="frontend.speech2mel.Speech2MelSpec", conf=dict(sr=22050, mag_spec=True, hop_length=256, win_length=1024, n_mels=80,
                                                                                fmin=0, fmax=8000, log_base=None, clamp=1e-5)),
        feat_normalize=True, pitch_normalize=True, energy_normalize=True,
        decoder=dict(type="transformer.encoder.TransformerEncoder", conf=dict(trf)),
        dec_postnet=dict(type="postnet.conv1d.Conv1dPostnet", conf=dict(conv_dims=[256, -1, -1, -1, 0], conv_kernel=5, conv_batchnorm=True,
                                                                         conv_activation="Tanh", conv_dropout=dropout_conv)))


def synth_tts(seed: int, device, nb: int = BATCH):
    """phoneme ids uniform in [1, 75), 128 tokens after the <sos/eos> strip, 163 584 samples at 22 050 Hz (640 frames at
    hop 256), 5 frames per token, frame-level pitch ~ N(0, 1)."""
    g = torch.Generator().manual_seed(2000 + seed)
    text = torch.randint(1, TTS_VOCAB, (nb, TTS_TOKENS), generator=g)
    text_len = torch.full((nb,), TTS_TOKENS, dtype=torch.int64)
    dur = torch.full((nb, TTS_TOKENS), 5.0)
    wave = torch.randn(nb, TTS_SAMPLES, 1, generator=g) * 0.1
    wave_len = torch.full((nb,), TTS_SAMPLES, dtype=torch.int64)
    pitch = torch.randn(nb, TTS_FRAMES, generator=g)
    pitch_len = torch.full((nb,), TTS_FRAMES, dtype=torch.int64)
    return [t.to(device) for t in (text, text_len, dur, text_len.clone(), wave, wave_len, pitch, pitch_len)]


class TTSTrainStepWorkload:
    """FastSpeech2 train step from waveforms: frontend (mel + energy) -> three global normalisations -> encoder -> variance predictors ->
    length regulation -> decoder -> feat_pred + Conv1d postnet -> five-term loss -> backward -> clip 1.0 + Adam (Noam, warm-up 6000).
    The head (frontend, normalisations with their running statistics, frame counts, per-token targets) runs eagerly; the rest of forward +
    backward is one hipGraph and clip + Adam (two launches over the flat buckets) another (one rank; several ranks run eagerly with the
    hook-launched bucket all-reduces).  Weight gradients through the grouped launch."""

    dtype = "bf16"
    name = ("FastSpeech2 train step (4enc/4dec d384 h2 conv-ffn1536 k3, predictors 256, postnet 256x4, bf16) from 32x7.4s@22.05kHz "
            "waveforms: frontend(mel+energy) + global normalisations + encoder + duration/pitch/energy predictors + length regulation + "
            "decoder + postnet + L1/L2 losses + backward + clip + Adam (BASELINE configs[4])")

    def __init__(self, device, seed, world, distributed=False):
        from speechain_amd import functional as Fn
        from speechain_amd.ddp import GradBucketReducer, attention_fusion_groups
        from speechain_amd.harness import TTSFastSpeech2
        from speechain_amd.optim import FusedAdam

        Fn.set_compute_dtype(torch.bfloat16)
        Fn.set_attention_probs(False)
        torch.manual_seed(0)
        self.model = TTSFastSpeech2(TTS_VOCAB, tts_recipe_conf(*TTS_DROPOUT)).to(device).train()
        if distributed and os.environ.get("BENCH_SYNCBN", "1") == "1":  # (the postnet's BatchNorm1d blocks: runner.py:1856-1858)
            self.model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(self.model, process_group=syncbn_group())
        for p in self.model.parameters():
            if p.dim() > 1:
                torch.nn.init.xavier_normal_(p)
        self.params = [p for p in self.model.parameters() if p.requires_grad]
        self.reducer = GradBucketReducer(self.params, bucket_mb=32.0, adjacent=attention_fusion_groups(self.model))
        self.reducer.broadcast_parameters()
        self.lr = torch.tensor([1e-3], device=device)
        self.noam_warmup = 6000
        self.opt = FusedAdam(self.reducer, self.lr, betas=(0.9, 0.98), eps=1e-9, max_grad_norm=1.0)
        torch.manual_seed(1234 + seed)
        self.batch = synth_tts(seed, device)
        self.nstep, self.world = 0, world
        self.ev, self.flops, self.launch_flops, self.launch_kernels = [], 0.0, [], []
        self.targets, self.g_body, self.g_tail, self.loss = None, None, None, None
        self.use_graph = os.environ.get("BENCH_NO_GRAPH", "0") != "1"
        self.head_stream = torch.cuda.Stream(device=device) if (os.environ.get("BENCH_NO_PREFETCH", "0") != "1" and self.use_graph) else None
        self.next_head, self.copied = None, None

    def _body(self):
        from speechain_amd import functional as Fn

        Fn.DropoutState.advance()
        Fn.weights.refresh()
        self.reducer.zero_grad()
        text, text_len = self.batch[0], self.batch[1]
        out = self.model(text, text_len, targets=self.targets)
        out["loss"].backward()
        self.reducer.settle()
        return out["loss"].detach()

    def step(self, timed: bool, instrument: bool = False):
        from speechain_amd._lib import lib

        self.nstep += 1
        self.lr.fill_(384 ** -0.5 * min(self.nstep ** -0.5, self.nstep * self.noam_warmup ** -1.5))
        # head, eager: frontend (mel + energy), the three normalisations (their running statistics are re-created on every call), frame counts per
        # token, per-token pitch / energy targets -> the static inputs of the captured body
        # Like the ASR workload's head, it depends on the data only: the head of batch i+1 runs on a second stream while the graphs of batch i
        # are replayed (every timed step still runs one head and one body).  BENCH_NO_PREFETCH=1: in line.
        main = torch.cuda.current_stream()
        if self.head_stream is None:
            tg = self.model.prepare(*self.batch[2:], epoch=1)
        else:
            if self.next_head is None:
                self._prefetch_next(main)
            tg, ev = self.next_head
            main.wait_event(ev)
        if self.targets is None:
            self.targets = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in tg.items()}
        else:
            with torch.no_grad():
                for k, v in tg.items():
                    if torch.is_tensor(v):
                        self.targets[k].copy_(v)
        if self.head_stream is not None:  # the head stream may reuse the memory of `tg` only after these copies have run
            self.copied = torch.cuda.Event()
            self.copied.record(main)
            self.next_head = None
        # (more than one rank: the body is replayed as one graph, then every bucket is all-reduced, then the optimizer graph -- no overlap of
        # the 35 M parameters' gradients with backward, but none of the eager path's launch overhead either: 16.2 -> 12.35 ms on a one-rank group)
        graph = self.use_graph and not instrument and self.nstep > 3
        if graph and self.g_body is None:
            self.reducer.wait()  # (collectives of the eager warm-up steps are complete before capture starts; see ASRTrainStepWorkload._capture)
            torch.cuda.synchronize()
            try:
                with no_gc():
                    self.g_body = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self.g_body, capture_error_mode="thread_local"):
                        self.loss = self._body()
                    self.g_tail = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self.g_tail, capture_error_mode="thread_local"):
                        self.opt.step()
            except Exception as e:  # (see ASRTrainStepWorkload.step)
                print(f"bench: capturing the step into hipGraphs failed ({type(e).__name__}: {e}); continuing with eager steps", file=sys.stderr)
                self.use_graph, graph = False, False
                self.g_body = self.g_tail = None
            torch.cuda.synchronize()
        if graph:
            self.g_body.replay()
            if self.reducer.multi:
                self.reducer.reduce_all_now()
            self.g_tail.replay()
            if self.head_stream is not None:
                self._prefetch_next(main)
            return self.loss
        if instrument:
            lib.timing = {"sc_gemm": self.ev, "sc_gemm_splitk": self.ev, "sc_wgrad_grouped": self.ev}
            lib.flop_log, lib.kernel_log = [], []
        loss = self._body()
        self.reducer.wait()
        self.opt.step()
        if instrument:
            self.flops += sum(lib.flop_log)
            self.launch_flops += lib.flop_log
            self.launch_kernels += lib.kernel_log
            lib.timing = lib.flop_log = lib.kernel_log = None
        return loss

    def _prefetch_next(self, main):
        with torch.cuda.stream(self.head_stream):
            if self.copied is not None:
                self.head_stream.wait_event(self.copied)
            else:
                self.head_stream.wait_stream(main)
            tg = self.model.prepare(*self.batch[2:], epoch=1)
            ev = torch.cuda.Event()
            ev.record(self.head_stream)
        self.next_head = (tg, ev)

    roofline = None  # bound below: the GEMM roofline code of the ASR workload works on the same logs


def cpu_baseline_tts(seconds: float):
    """The oracle's FastSpeech2 step (forward + backward, no optimizer) on the host cores over 4 of the 32 utterances."""
    from oracle import cpu_ref as R
    from speechain_amd.harness import TTSFastSpeech2

    cores = min(os.cpu_count() or 1, int(os.environ.get("BENCH_CPU_THREADS", 16)))
    torch.set_num_threads(cores)
    nb = 4
    torch.manual_seed(0)
    model = TTSFastSpeech2(TTS_VOCAB, tts_recipe_conf())
    P = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and "running" not in k and not k.endswith("posenc") and
                                              "mel_fbanks" not in k and "preemph" not in k) for k, v in model.state_dict().items()}
    cfg = dict(d=384, heads=2, layers=4, dur_layers=2, pitch_layers=5, energy_layers=2, post_layers=5)
    text, text_len, dur, _, wave, wave_len, pitch, pitch_len = synth_tts(0, "cpu", nb)
    norms = {k: R.FeatureNormalization() for k in ("feat", "pitch", "energy")}

    def one():
        out = R.fastspeech2_forward_loss(text, text_len, dur, wave, wave_len, pitch, pitch_len, P, cfg, norms, epoch=1)
        out["loss"].backward()

    one()
    t0, n = time.perf_counter(), 0
    while True:
        one()
        n += 1
        if time.perf_counter() - t0 > seconds or n >= 50:
            break
    dt = time.perf_counter() - t0
    return dict(value=round(nb * n / dt, 4), unit="utterances/s", cores=cores, kind="port",
                sample=f"{n} FastSpeech2 steps (fwd+bwd, no optimizer) over {nb} of the 32 utterances each, torch-CPU oracle, fp32, {cores} threads, {dt:.1f} s")


DECODE_CONF = dict(beam_size=16, ctc_weight=0.2)  # infer_cfg of the transformer recipes (100-bpe5k_transformer-large_lr2e-3.yaml:89-94)


class ASRDecodeWorkload:
    """the reference's ARASR.inference step 1 (model/ar_asr.py:854-878) on one batch -- 12-layer encoder over the
    (32, 249, 512) prenet output, then beam search over the 6-layer decoder with joint CTC scoring, beam 16, CTC weight 0.2, length
    limit 249 / 3 = 83 tokens.  Random weights never emit <sos/eos> early, so every sentence runs to the limit: 82 decoder steps of 512
    hypothesis rows.  Cached keys / values, CTC prefix scores and the ancestry tables are HIP kernels; the beam bookkeeping is host code
    fed by one device-to-host copy per step."""

    dtype = "bf16"
    name = ("ASR beam-search decoding (12enc/6dec d512 h8 ffn2048, bpe5k, bf16): encoder + beam search, beam 16, joint CTC prefix "
            "scoring weight 0.2, 82 cached-KV decoder steps of 512 hypothesis rows per 32-utterance batch (SURVEY f3)")

    def __init__(self, device, seed):
        from speechain_amd import functional as Fn
        from speechain_amd.harness import ASRTransformer

        Fn.set_compute_dtype(torch.bfloat16)
        Fn.set_attention_probs(False)
        torch.manual_seed(0)
        self.model = ASRTransformer(**ASR_MODEL).to(device).eval()
        g = torch.Generator().manual_seed(77 + seed)
        self.feat = torch.randn(BATCH, 249, 512, generator=g).to(device)
        self.feat_len = torch.full((BATCH,), 249, dtype=torch.int64, device=device)
        self.last = None

    def step(self, timed: bool, instrument: bool = False):
        from speechain_amd._lib import lib

        if instrument:
            names = ("sc_ctc_prefix_scores", "sc_ctc_prefix_scores_fast", "sc_ctc_prefix_state", "sc_attention_decode", "sc_attention_fwd", "sc_gemm",
                     "sc_layernorm_fwd")
            lib.timing = {k: [] for k in names}
        from speechain_amd.infer_func import beam_search as BS

        BS.REPLAY_GRAPHS = not instrument  # (events around launches see nothing inside a graph replay)
        self.last = self.model.inference(dict(DECODE_CONF), self.feat, self.feat_len)
        BS.REPLAY_GRAPHS = True
        if instrument:
            torch.cuda.synchronize()
            self.kernel_ms = {k: (len(v), sum(a.elapsed_time(b) for a, b in v)) for k, v in lib.timing.items()}
            lib.timing = None
        return self.last

    def roofline(self):
        """The CTC prefix-score kernel: per launch it reads the (B, T, V) f32 factor tensor E once (the 16 beams of a sentence are one
        matrix-core tile of rows) and writes psi (N, V): byte-bound -- the multiply-add rate is given beside it."""
        fast = self.kernel_ms.get("sc_ctc_prefix_scores_fast", (0, 0.0))[0] > 0  # (the factorised form: the default)
        n, ms = self.kernel_ms["sc_ctc_prefix_scores_fast" if fast else "sc_ctc_prefix_scores"]
        per = ms / max(n, 1)
        T, V, N = 249, ASR_MODEL["vocab_size"], BATCH * DECODE_CONF["beam_size"]
        alg = BATCH * T * V * 4 + N * V * 4 + T * 2 * N * 4
        achieved = alg / (per * 1e-3) / 1e9
        rate = N * V * (T - 1) / (per * 1e-3) / 1e12
        mfma = fast and V % 4 == 0  # (exact-f32 matrix-core products, the 16 rows of a beam = one tile)
        return {"bound": "hbm", "kernel": "ctc_psi_mfma_kernel" if mfma else "ctc_psi_fast_kernel" if fast else "ctc_psi_kernel",
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": pmc_traffic("ctc_psi_mfma_kernel<true>") if mfma else None,
                "kernel_ms": round(per, 5), "algorithmic_bytes": alg,
                "note": (f"{rate:.2f} T multiply-adds/s: one per (row, token, frame), the exponential's second factor made once per batch; "
                         if fast else f"{rate:.2f} T exp/s: one v_exp per (row, token, frame); ") + "launches per batch "
                        + ", ".join(f"{k} {c} x {t / max(c, 1) * 1e3:.1f} us" for k, (c, t) in self.kernel_ms.items() if c)}


def cpu_baseline_decode(seconds: float):
    """The oracle's beam search (decoder re-run on the whole prefix at every step, (T, 2, N, V) CTC tensor: the reference's algorithm) on
    the host cores for ONE of the 32 utterances, encoder included, cut after as many decoding steps as fit ~`seconds`."""
    from oracle import cpu_ref as R
    from oracle import decode_ref as D
    from speechain_amd.harness import ASRTransformer

    cores = min(os.cpu_count() or 1, int(os.environ.get("BENCH_CPU_THREADS", 16)))
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    model = ASRTransformer(**ASR_MODEL)
    P = {k: v.detach().clone() for k, v in model.state_dict().items()}
    P["embed.weight"] = P.pop("embed.embed.weight")
    cfg = dict(enc_layers=12, dec_layers=6, num_heads=8, d_model=512, activation="GELU", emb_scale=False)
    g = torch.Generator().manual_seed(77)
    feat = torch.randn(1, 249, 512, generator=g)
    mask = R.make_mask_from_len(torch.tensor([249]))
    steps = {"n": 0, "t0": 0.0}

    class Cut(Exception):
        pass

    def decode_fn(e, m, t, tl):
        if steps["n"] >= 3 and time.perf_counter() - steps["t0"] > seconds:
            raise Cut()
        steps["n"] += 1
        return D.asr_decode_logits(e, m, t, tl, P, cfg)

    with torch.no_grad():
        t_all = time.perf_counter()
        enc, mask, _, _ = R.transformer_encoder(feat, mask, P, "encoder.", num_layers=12, num_heads=8, d_model=512, activation="GELU")
        steps["t0"] = time.perf_counter()
        try:
            D.beam_search(enc, mask, decode_fn, 5000, sos_eos=4999, padding_idx=0, ctc_logits_fn=lambda e: R.token_postnet(e, P, "ctc_layer."),
                          **DECODE_CONF)
        except Cut:
            pass
        dt = time.perf_counter() - t_all
    # a full utterance takes 82 steps whose cost grows with the prefix; the measured steps are the cheapest ones, so this is an upper bound
    est = (steps["t0"] - t_all) + (time.perf_counter() - steps["t0"]) * 82 / max(steps["n"], 1)
    return dict(value=round(1.0 / est, 5), unit="utterances/s", cores=cores, kind="port",
                sample=f"encoder + the first {steps['n']} of 82 beam-search steps of 1 of the 32 utterances (beam 16, CTC 0.2; torch-CPU "
                       f"oracle, fp32, {cores} threads, {dt:.1f} s), scaled to 82 steps at the measured per-step cost (an upper bound on the "
                       f"rate: later steps decode longer prefixes)")


def pmc_traffic(kernel: str):
    """HBM bytes per launch of `kernel` as measured by the committed PMC passes (None when there is no such record)."""
    for name in ("r03_frontend_pmc_traffic.json", "r03_train_step_pmc_traffic.json", "r03_decode_pmc_traffic.json", "r02_frontend_pmc_traffic.json", "r02_train_step_pmc_traffic.json",
                 "r01_frontend_pmc_traffic.json", "r01_train_step_pmc_traffic.json"):
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", name)
        try:
            with open(path) as f:
                return json.load(f)[kernel]["hbm_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            continue
    return None


def pmc_mfma_busy(kernel: str):
    """Matrix-core busy fraction of `kernel` from the committed SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE pass (None when there is none)."""
    for name in ("r03_train_step_pmc_mfma.json", "r02_train_step_pmc_mfma.json"):
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", name)
        try:
            with open(path) as f:
                return json.load(f)[kernel]["mfma_busy_frac"]
        except (OSError, KeyError, ValueError):
            continue
    return None


def cpu_baseline_train(seconds: float, prenet_dims=(64, 64)):
    """The oracle's ASR step (frontend -> normalisation -> specaug -> prenet -> encoder -> decoder -> losses -> backward) on the host cores,
    on a bounded sample: ~`seconds` of steps over 4 of the 32 utterances (same per-utterance shapes), scaled to utterances/s."""
    from oracle import cpu_ref as R
    from speechain_amd.harness import ASRTransformer
    from speechain_amd.module.prenet.conv2d import Conv2dPrenet

    cores = min(os.cpu_count() or 1, int(os.environ.get("BENCH_CPU_THREADS", 16)))
    torch.set_num_threads(cores)
    nb = 4
    torch.manual_seed(0)
    pconf = prenet_conf(prenet_dims)
    # (the modules only supply initial weights here; the arithmetic below is the oracle's)
    PP = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and "running" not in k)
          for k, v in Conv2dPrenet(input_size=80, **pconf).state_dict().items()}
    model = ASRTransformer(**ASR_MODEL)
    P = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and not k.endswith("posenc")) for k, v in model.state_dict().items()}
    P["embed.weight"] = P.pop("embed.embed.weight")
    cfg = dict(enc_layers=12, dec_layers=6, num_heads=8, d_model=512, activation="GELU", label_smoothing=0.2, ctc_weight=0.3)
    wave, _, lens = synth_batch(0, "cpu")
    text, text_len = synth_text(0, "cpu")
    wave, lens, text, text_len = wave[:nb].squeeze(-1), lens[:nb], text[:nb], text_len[:nb]

    norm = R.FeatureNormalization()

    def one():
        feat, flen = R.speech2mel(wave, lens, **ASR_FRONTEND)
        feat, flen = norm(feat, flen, epoch=1)
        d = R.specaug_draw(flen, nb, 80, **SPECAUG)
        feat = R.specaug_apply(feat, flen, d)
        x, xlen, _ = R.conv2d_prenet(feat, flen, PP, "", **pconf, training=True)
        out = R.asr_forward_loss(x, xlen, text, text_len, P, cfg)
        out["loss"].backward()

    one()  # warm-up
    t0 = time.perf_counter()
    n = 0
    while True:
        one()
        n += 1
        if time.perf_counter() - t0 > seconds or n >= 50:
            break
    dt = time.perf_counter() - t0
    return dict(value=round(nb * n / dt, 4), unit="utterances/s", cores=cores, kind="port",
                sample=f"{n} train steps (fwd+bwd, no optimizer) over {nb} of the 32 utterances each, torch-CPU oracle, fp32, "
                       f"{cores} threads, {dt:.1f} s")


def cpu_baseline(seconds: float):
    """The oracle's frontend + SpecAugment on the host cores, same batch shape; bounded to ~`seconds` of work."""
    from oracle import cpu_ref as R  # test infrastructure used as the timed CPU port

    # the GPU box gives one GPU's job a 16-core share of the host; more torch threads than that only thrash
    cores = min(os.cpu_count() or 1, int(os.environ.get("BENCH_CPU_THREADS", 16)))
    torch.set_num_threads(cores)
    wave, _, lens = synth_batch(0, "cpu")
    wave = wave.squeeze(-1)

    def one():
        feat, flen = R.speech2mel(wave, lens, **ASR_FRONTEND)
        d = R.specaug_draw(flen, BATCH, 80, **SPECAUG)
        return R.specaug_apply(feat, flen, d)

    one()
    one()
    t0 = time.perf_counter()
    n = 0
    while True:
        one()
        n += 1
        if time.perf_counter() - t0 > seconds or n >= 200:
            break
    dt = time.perf_counter() - t0
    return dict(value=round(BATCH * n / dt, 2), unit="utterances/s", cores=cores, kind="port",
                sample=f"{n} batches of 32x160000 samples (frontend+specaug, torch-CPU oracle, fp32, {cores} threads)")


def self_launch(args) -> int:
    """`python bench.py --gpus N` with no launcher around it (no RANK / WORLD_SIZE in the environment): start the N ranks here, one child
    process per GPU, as the reference's runner does with mp.spawn (speechain/runner.py:2014).  The parent never touches the GPU (children
    are fresh interpreters started before any HIP call -- never an exec of a process that initialised the device); it relays rank 0's one
    JSON line and returns non-zero if any rank failed (the other ranks are then ended, by the PIDs started here)."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    n = args.gpus
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    import threading

    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)  # (drain rank 0's pipe while polling all ranks)
    reader.start()
    rc = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench: rank {r} exited with code {code}; ending the other ranks", file=sys.stderr)
                for q in alive:
                    procs[q].terminate()
        time.sleep(0.05)
    reader.join(timeout=10)
    if rc == 0:
        sys.stdout.write(out0[0] if out0 else "")
        sys.stdout.flush()
    return rc


def launch_check(args, rank, local_rank, world):
    """--launch-check: process group up, one all-reduce, one line."""
    import torch.distributed as dist

    gpu = torch.cuda.is_available()
    if gpu:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank) if gpu else torch.device("cpu")
    if world > 1:
        dist.init_process_group(backend="nccl" if gpu else "gloo", **({"device_id": dev} if gpu else {}))
    t = torch.tensor([float(rank + 1)], device=dev)
    if world > 1:
        dist.all_reduce(t)
        dist.barrier()
    ok = float(t.item()) == world * (world + 1) / 2
    return {"launch_check": "ok" if ok else "sum mismatch", "world": world, "backend": ("nccl" if gpu else "gloo") if world > 1 else None,
            "sum": float(t.item())}


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    # stdout carries exactly ONE line, the JSON result of rank 0.  Libraries print there too (RCCL writes a version banner to
    # stdout when the process group comes up), so file descriptor 1 points at stderr until that line is written.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run, or with no launcher at all)"
    if args.launch_check:
        res = launch_check(args, rank, local_rank, world)
        if rank == 0:
            sys.stdout.flush()
            os.dup2(real_stdout, 1)
            print(json.dumps(res), flush=True)
            os.dup2(2, 1)
        if world > 1:
            torch.distributed.destroy_process_group()
        sys.exit(0 if res["launch_check"] == "ok" else 1)
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    # BENCH_FORCE_DIST=1 (with RANK/WORLD_SIZE/MASTER_* from torch.distributed.run --nproc-per-node 1) runs the multi-rank code
    # path -- RCCL process group, parameter broadcast, bucket all-reduces between the two hipGraphs, barriers, MAX over ranks --
    # on a one-rank group: the rehearsal of the N > 1 launch that a single-GPU box allows.
    force_dist = os.environ.get("BENCH_FORCE_DIST", "0") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist

        if force_dist:
            os.environ["SC_DDP_FORCE_COLLECTIVES"] = "1"
        dist.init_process_group(backend="nccl", device_id=device)

    torch.manual_seed(1234 + rank)
    prenet_dims = tuple(int(c) for c in args.prenet_dims.split(","))
    if args.workload == "frontend":
        wl = FrontendWorkload(device, seed=rank)
    elif args.workload == "asr_decode":
        wl = ASRDecodeWorkload(device, rank)
    elif args.workload == "tts_train_step":
        TTSTrainStepWorkload.roofline = ASRTrainStepWorkload.roofline
        wl = TTSTrainStepWorkload(device, rank, world, distributed=dist is not None)
    else:
        wl = ASRTrainStepWorkload(device, rank, world, prenet_dims, distributed=dist is not None)
    for _ in range(args.warmup):
        wl.step(False)
    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # A full (generation-2) pass of Python's cyclic collector over the heap this process has built by now takes ~40 ms, and the allocation
    # counters reach its threshold some tens of steps into the timed region (seen as one 40 ms step in the 0.08 ms frontend workload): collect
    # now and move the survivors out of the collector's reach, as a long-running training process would at start-up.
    gc.collect()
    gc.freeze()
    fence()
    t0 = time.perf_counter()
    dbg = [] if os.environ.get("BENCH_DEBUG_STEPS") else None  # (host time of every timed step and of the final drain, to stderr)
    for i in range(args.steps):
        wl.step(True)  # every timed step is the same replayed step
        if dbg is not None:
            dbg.append(time.perf_counter())
    fence()
    elapsed = time.perf_counter() - t0
    if dbg is not None and rank == 0:
        print("host us per timed step:", [round(1e6 * (b - a)) for a, b in zip([t0] + dbg[:-1], dbg)], "drain us:", round(1e6 * (t0 + elapsed - dbg[-1])), file=sys.stderr)
    if args.workload in ("asr_train_step", "tts_train_step", "asr_decode"):
        # one more step, eager, with HIP events around every GEMM launch: the roofline's per-kernel times.  It runs right AFTER the timed
        # region (it takes about twice a replayed step and, on several ranks, uses the hook-launched all-reduce path), in the same
        # process, on the same weights and inputs; every rank runs it (its collectives need all of them), rank 0 reports its events.
        wl.step(False, instrument=True)
        fence()
    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        if args.workload == "frontend":
            kms = wl.dominant_ms()
            achieved = wl.alg_bytes / (kms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": wl.dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": pmc_traffic(wl.dominant),
                    "kernel_ms": round(kms, 5), "algorithmic_bytes": wl.alg_bytes,
                    "traffic_note": "HBM bytes per launch from the committed rocprofv3 FETCH_SIZE/WRITE_SIZE passes "
                                    "(profiles/r03_frontend_pmc_traffic.json, or the newest earlier round's); bench.py cannot run the profiler on itself"}
        else:
            roof = wl.roofline()
        line = {
            "metric": ("utterances/sec (7.4 s@22.05 kHz, bsz32) FastSpeech2 train step" if args.workload == "tts_train_step" else
                       "utterances/sec (10 s@16 kHz, bsz32) beam-search decoding, beam 16, CTC weight 0.2" if args.workload == "asr_decode" else
                       "utterances/sec (10 s@16 kHz, bsz32)"),
            "value": round(BATCH * args.steps * world / elapsed, 2),
            "unit": "utterances/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": wl.dtype,
            "data": "synthetic",
            "config": {"workload": wl.name, "per_gpu_batch": BATCH, "samples_per_utt": TTS_SAMPLES if args.workload == "tts_train_step" else SAMPLES,
                       "sharding": (f"{world} independent per-GPU batches, no collective" if args.workload in ("frontend", "asr_decode") else
                                    f"dp{world}: per-GPU batch 32; body hipGraph, then the gradient all-reduce in 32 MB flat buckets on a "
                                    "communication stream, then the optimizer hipGraph (a single rank: no collective)" if args.workload == "tts_train_step" else
                                    f"dp{world}: per-GPU batch 32; one exchange per step, the RCCL gradient all-reduce in 32 MB flat "
                                    "buckets" + (" -- backward is replayed as five hipGraphs (cut at the encoder output, two encoder "
                                    "hidden states and the encoder input) and the buckets each piece completes are reduced on a "
                                    "communication stream while the next piece runs; the optimizer hipGraph waits for them"
                                                 if getattr(wl, "segmented", False) else
                                                 " (a single rank: no collective is issued; forward + backward is one hipGraph)")
                                    + "; plus the two (80,)-vector all-gathers of the feature "
                                    "normalisation statistics in the eager part of the step, as in the reference")},
            "roofline": roof,
        }
        if args.workload in ("asr_train_step", "tts_train_step"):
            line["config"]["hipgraphs"] = bool(getattr(wl, "use_graph", False) and wl.g_body is not None)  # (False: BENCH_NO_GRAPH=1 or a failed capture)
            line["config"]["batchnorm"] = ("SyncBatchNorm statistics over all ranks (speechain/runner.py:1856-1858), (C,)-sized all-reduces on a "
                                           "process group of their own" if (world > 1 and os.environ.get("BENCH_SYNCBN", "1") == "1") else "per-rank batch statistics (one rank)")
        if args.workload == "asr_train_step":
            line["config"]["prenet"] = PRENET_NOTE
            step_s = elapsed / args.steps
            tf = NAMED_PATH_FLOPS_PER_STEP / step_s / 1e12
            roof["step_named_path"] = {"flops_per_step": NAMED_PATH_FLOPS_PER_STEP, "achieved": round(tf, 2),
                                       "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 5),
                                       "note": "SURVEY.md section 8d config 3: 848 GFLOP forward x 3 of the named path (encoder, decoder, "
                                               "heads; prenet excluded) per 32-utterance step / measured step time / dense bf16 peak"}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = (cpu_baseline(args.cpu_seconds) if args.workload == "frontend" else
                                    cpu_baseline_decode(args.cpu_seconds) if args.workload == "asr_decode" else
                                    cpu_baseline_tts(args.cpu_seconds) if args.workload == "tts_train_step" else
                                    cpu_baseline_train(args.cpu_seconds, prenet_dims))
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
