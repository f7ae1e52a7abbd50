"""CPU, world_size 2 over gloo: the training-time exchange of the replicas (u2tokenizer_amd/dp.py, ZeRO-1 semantics of
config/ds_config.json:27-39).  Two ranks with DIFFERENT micro-batches must end every step with identical parameters,
equal to a single process running torch.optim.AdamW on the MEAN gradient; buckets smaller than the model force several
reduce-scatter / all-gather rounds (a bucket is cut into world_size pieces regardless of parameter boundaries)."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _WithDead(torch.nn.Sequential):
    """the plain model + a Linear that the forward never touches -- unless `wake` is set (a data-dependent branch: a
    parameter without gradient for some steps, then with one)"""

    def __init__(self, *mods):
        super().__init__(*mods)
        self.dead = torch.nn.Linear(24, 24)
        self.wake = False

    def forward(self, x):
        if self.wake:   # at the INPUT side: its gradient is the last of the backward, long after its bucket (filled first,
            x = x + self.dead(x)   # with the last layers) has been reduced from the learnt subset
        for name, mod in self.named_children():
            if name != "dead":
                x = mod(x)
        return x


def _model(variant="plain"):
    torch.manual_seed(0)
    mods = [torch.nn.Linear(24, 40), torch.nn.GELU(), torch.nn.Linear(40, 40), torch.nn.LayerNorm(40), torch.nn.Linear(40, 7)]
    return _WithDead(*mods) if variant in ("unused", "wakes", "wakes1") else torch.nn.Sequential(*mods)


def _steps(variant):
    return 12 if variant == "wakes1" else 8 if variant in ("unused", "wakes") else 3


def _wake(m, variant, step, rank=0):
    """variant "wakes": the dead Linear takes part from step 5 on -- after its bucket has been launched early from the learnt
    subset for a step or two.  "wakes1": ONLY RANK 1 sees it, in steps 5 and 6 (per-rank data: a text-only batch on the other
    rank) -- ADVICE r4: what a rank does about it must not depend on what that rank alone has seen."""
    if variant == "wakes":
        m.wake = step >= 5
    elif variant == "wakes1":
        m.wake = rank == 1 and step in (5, 6)


def _data(rank, step):
    g = torch.Generator().manual_seed(100 * step + rank)
    return torch.randn(5, 24, generator=g), torch.randn(5, 7, generator=g)


def _worker(rank, world, port, q, overlap, bucket, clip=None, accum=1, variant="plain"):
    """variant: "plain"; "unused" = a parameter that never receives a gradient sits in the model (linear_aggregator.wv / dense
    of the reference, tta.py:47-48,62-65); "groups" = HF-style parameter groups (no decay on biases / LayerNorm); "none" = the
    caller clears gradients with set_to_none=True (what HF Trainer's model.zero_grad() does)."""
    import traceback
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from u2tokenizer_amd.dp import Zero1AdamW, hf_param_groups
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = _model(variant)
        params = hf_param_groups(m, 0.1) if variant == "groups" else m.parameters()
        opt = Zero1AdamW(params, lr=1e-2, weight_decay=0.1, reduce_bucket_size=bucket, allgather_bucket_size=bucket,
                         overlap_comm=overlap, max_grad_norm=clip, gradient_accumulation_steps=accum)
        norms, launched_early = [], []
        for step in range(_steps(variant)):
            _wake(m, variant, step, rank)
            for micro in range(accum):
                x, y = _data(rank, step * accum + micro)
                (torch.nn.functional.mse_loss(m(x), y) / accum).backward()
                if micro < accum - 1 and variant not in ("wakes", "wakes1"):
                    assert opt._next_launch == 0, "a bucket was reduced before the last micro-batch"
            launched_early.append(opt._next_launch)   # buckets whose reduce-scatter was launched from a hook (overlap)
            opt.step()
            norms.append(opt.last_grad_norm)
            if variant == "none":
                m.zero_grad(set_to_none=True)
            else:
                opt.zero_grad()
        # numpy arrays travel through the queue BY VALUE; torch tensors would travel as shared-memory file descriptors that
        # the parent has to fetch from this process while it is still alive (ConnectionResetError on a loaded host)
        q.put((rank, [p.detach().clone().numpy() for p in m.parameters()], len(opt.buckets), opt.state_bytes_per_rank(), norms,
               launched_early))
    except Exception:  # a failure of the exchange under test must surface as itself, not as a start-up hiccup
        q.put((rank, "error", traceback.format_exc()))
    dist.barrier()
    dist.destroy_process_group()


@torch.enable_grad()  # (other test modules of the suite switch autograd off process-wide at import)
def _reference(world, clip=None, accum=1, variant="plain"):
    from u2tokenizer_amd.dp import hf_param_groups
    m = _model(variant)
    opt = torch.optim.AdamW(hf_param_groups(m, 0.1) if variant == "groups" else m.parameters(), lr=1e-2, weight_decay=0.1)
    norms = []
    for step in range(_steps(variant)):
        opt.zero_grad()
        woke = False
        for rank in range(world):
            _wake(m, variant, step, rank)
            woke = woke or bool(getattr(m, "wake", False))
            for micro in range(accum):
                x, y = _data(rank, step * accum + micro)
                (torch.nn.functional.mse_loss(m(x), y) / (world * accum)).backward()
        if variant in ("unused", "wakes", "wakes1") and not woke:
            # a flat ZeRO partition has a (zero) gradient for every element, so AdamW's decoupled weight decay also shrinks
            # parameters that received none -- DeepSpeed's behaviour, unlike torch.optim.AdamW skipping grad-less tensors
            for p in m.dead.parameters():
                p.grad = torch.zeros_like(p)
        if clip is not None:
            norms.append(float(torch.nn.utils.clip_grad_norm_(m.parameters(), clip)))
        opt.step()
    return [p.detach().clone() for p in m.parameters()], norms


def _run(overlap, bucket, clip=None, accum=1, variant="plain"):
    # one retry, and only for what is the launcher's environment rather than the exchange under test: the result queue of a
    # spawned pair breaking on a loaded host (EOFError / ConnectionError while unpickling shared-memory tensors).  A worker
    # that raises reports its traceback through the queue and FAILS the test; so does a non-zero exit code.
    try:
        return _run_once(overlap, bucket, clip, accum, variant)
    except (EOFError, ConnectionError, FileNotFoundError):
        return _run_once(overlap, bucket, clip, accum, variant)


def _run_once(overlap, bucket, clip=None, accum=1, variant="plain"):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, overlap, bucket, clip, accum, variant)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[1] != "error", r[2]
    res = [(r[0], [torch.from_numpy(a) for a in r[1]], *r[2:]) for r in res]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    return res


def test_zero1_matches_single_process_adamw():
    ref, _ = _reference(2)
    nparam = sum(p.numel() for p in ref)
    for overlap, bucket in ((False, 10 ** 9), (True, 700), (True, 10 ** 9)):
        (r0, p0, nb0, sb0, _, early), (r1, p1, nb1, sb1, _, _) = _run(overlap, bucket)
        assert all(e == (nb0 if overlap else 0) for e in early)      # with overlap_comm every bucket goes from its hook
        for a, b, c in zip(p0, p1, ref):
            assert torch.equal(a, b)                                  # replicas stay bit-identical
            assert torch.allclose(a, c, rtol=2e-5, atol=2e-6), (a - c).abs().max()
        assert nb0 == nb1 and (nb0 > 1) == (bucket == 700)
        # the optimiser state is sharded: ~half of 12 bytes per parameter on each of the two ranks
        assert abs(sb0 - 6 * nparam) <= 12 * 2 * nb0 and sb0 == sb1


def test_zero1_global_gradient_clipping():
    """max_grad_norm (ds_config.json:41 "gradient_clipping"): the norm of the MEAN gradient over all parameters is assembled
    from the ranks' bucket pieces (one scalar all-reduce) and equals torch.nn.utils.clip_grad_norm_ on a single process; the
    clipped steps agree and the replicas stay bit-identical."""
    clip = 0.05                                        # well below the actual norms: every step is clipped
    ref, ref_norms = _reference(2, clip)
    assert min(ref_norms) > 2 * clip
    (r0, p0, _, _, n0, _), (r1, p1, _, _, n1, _) = _run(True, 700, clip)
    assert n0 == n1
    for a, b in zip(n0, ref_norms):
        assert abs(a - b) <= 1e-5 * b
    for a, b, c in zip(p0, p1, ref):
        assert torch.equal(a, b)
        assert torch.allclose(a, c, rtol=2e-5, atol=2e-6), (a - c).abs().max()


def test_zero1_gradient_accumulation_unused_parameters_and_groups():
    """config/ds_config.json:40 gradient_accumulation_steps: the reduce-scatter of a bucket must wait for the LAST micro-batch
    (the worker asserts nothing was launched earlier), gradients accumulate in the flat buckets, and the result equals one
    process accumulating world x accum micro-batches.  A parameter that never receives a gradient costs the overlap only in
    the first three steps (then the subset of parameters that do is trusted); one that starts receiving gradients later
    ("wakes": ADVICE r3 -- the early launch of that step is repeated by step(), nothing raises, parameters still equal AdamW's).  HF-style parameter groups (no decay on biases / LayerNorm)
    and a caller that clears gradients with set_to_none=True give the same parameters as torch.optim.AdamW."""
    for variant, accum, clip in (("plain", 3, None), ("unused", 2, 0.05), ("groups", 1, None), ("none", 2, None),
                                 ("wakes", 1, None), ("wakes", 2, 0.05)):
        ref, ref_norms = _reference(2, clip, accum, variant)
        (r0, p0, nb, _, n0, early), (r1, p1, _, _, n1, _) = _run(True, 700, clip, accum, variant)
        assert n0 == n1
        for a, b, c in zip(p0, p1, ref):
            assert torch.equal(a, b), variant
            assert torch.allclose(a, c, rtol=2e-5, atol=2e-6), (variant, (a - c).abs().max())
        if clip is not None:
            for a, b in zip(n0, ref_norms):
                assert abs(a - b) <= 1e-5 * b
        if variant == "unused":   # the subset is trusted once it has been the same for 3 steps
            assert all(e < nb for e in early[:3]) and all(e == nb for e in early[3:]), early
        elif variant == "wakes":
            # ... and forgotten when the parameter wakes up in step 5.  One micro-batch per step: its bucket had been launched
            # from the learnt subset by then, step() reduces it again.  Two: the first micro-batch already shows the
            # stranger, the bucket (and, in order, every bucket behind it) waits for step().
            assert all(e < nb for e in early[:3]) and early[3] == early[4] == early[6] == early[7] == nb, early
            assert early[5] == (nb if accum == 1 else 0), early
        else:
            assert all(e == nb for e in early), (variant, early)


def test_zero1_one_rank_alone_sees_the_late_gradient():
    """the learnt "which parameters of a bucket fire" gates early launches, and whether a premature launch must be
    repeated is a COLLECTIVE decision -- so neither may depend on what one rank alone has seen.  Here only rank 1's batches
    reach the dead Linear, in steps 5 and 6: in step 5 its bucket has already left from the learnt subset on both ranks (rank 1
    notices afterwards: every rank reduces it again), in step 6 rank 1's hooks complete the bucket while rank 0 waits for
    step(); then the subset is learnt again, by both ranks in the same step.  The collective sequences stay paired (no hang,
    no garbage), the replicas stay bit-identical and equal to one process running AdamW on the mean gradient."""
    for accum, clip in ((1, None), (2, 0.05)):
        ref, ref_norms = _reference(2, clip, accum, "wakes1")
        (r0, p0, nb, _, n0, e0), (r1, p1, _, _, n1, e1) = _run(True, 700, clip, accum, "wakes1")
        assert n0 == n1
        for a, b, c in zip(p0, p1, ref):
            assert torch.equal(a, b)
            assert torch.allclose(a, c, rtol=2e-5, atol=2e-6), (a - c).abs().max()
        if clip is not None:
            for a, b in zip(n0, ref_norms):
                assert abs(a - b) <= 1e-5 * b
        # rank 0 never sees the stranger: learnt after 3 steps, forgotten with rank 1 after step 5, learnt again after 7..9
        assert all(e < nb for e in e0[:3]) and e0[3] == e0[4] == e0[5] == nb, e0
        assert all(e < nb for e in e0[6:10]) and e0[10] == e0[11] == nb, e0
        # rank 1: as rank 0, except that in step 6 (all of its parameters fire, subset forgotten) every bucket goes from a hook
        assert all(e < nb for e in e1[:3]) and e1[3] == e1[4] == nb and e1[6] == nb, e1
        assert e1[5] == (nb if accum == 1 else 0), e1
        assert all(e < nb for e in e1[7:10]) and e1[10] == e1[11] == nb, e1


@torch.enable_grad()
def test_single_process_zero1_is_plain_adamw():
    from u2tokenizer_amd.dp import Zero1AdamW
    m1, m2 = _model(), _model()
    o1, o2 = Zero1AdamW(m1.parameters(), lr=1e-2, weight_decay=0.1), torch.optim.AdamW(m2.parameters(), lr=1e-2, weight_decay=0.1)
    for step in range(3):
        x, y = _data(0, step)
        for m, o in ((m1, o1), (m2, o2)):
            o.zero_grad()
            torch.nn.functional.mse_loss(m(x), y).backward()
            o.step()
    for a, b in zip(m1.parameters(), m2.parameters()):
        assert torch.allclose(a, b, rtol=2e-5, atol=2e-6)


@torch.enable_grad()
def test_zero1_mixed_dtype_buckets_keep_their_precision():
    """Buckets are split by dtype; each kind has its own staging buffers for the updated parameters (one buffer of
    the first bucket's dtype rounded fp32 parameters through bf16 on every step)."""
    from u2tokenizer_amd.dp import Zero1AdamW
    torch.manual_seed(0)
    a16, b32 = torch.nn.Linear(6, 5).to(torch.bfloat16), torch.nn.Linear(6, 5)
    ref = torch.nn.Linear(6, 5)
    ref.load_state_dict(b32.state_dict())
    opt = Zero1AdamW(list(a16.parameters()) + list(b32.parameters()), lr=1e-2, weight_decay=0.1)
    ropt = torch.optim.AdamW(ref.parameters(), lr=1e-2, weight_decay=0.1)
    assert len({b.flat_grad.dtype for b in opt.buckets}) == 2
    for step in range(3):
        x = torch.randn(4, 6, generator=torch.Generator().manual_seed(step))
        (a16(x.to(torch.bfloat16)).float().square().mean() + b32(x).square().mean()).backward()
        ref(x).square().mean().backward()
        opt.step(); opt.zero_grad(); ropt.step(); ropt.zero_grad()
    for p, q in zip(b32.parameters(), ref.parameters()):
        assert torch.allclose(p, q, rtol=2e-5, atol=2e-6)
        assert not torch.equal(p, p.to(torch.bfloat16).float())      # still full fp32 values
    assert all(p.dtype == torch.bfloat16 for p in a16.parameters())


@torch.enable_grad()
def test_zero1_state_dict_resumes_bit_identically():
    """Optimiser shard save / load (checkpoint-resume of a training run): two steps, save, two more steps == two steps, save,
    fresh optimiser + load, two more steps -- bit for bit; a shard of another layout is refused."""
    import pytest
    from u2tokenizer_amd.dp import Zero1AdamW

    def steps(m, o, lo, hi):
        for step in range(lo, hi):
            x, y = _data(0, step)
            o.zero_grad()
            torch.nn.functional.mse_loss(m(x), y).backward()
            o.step()

    m1 = _model()
    o1 = Zero1AdamW(m1.parameters(), lr=1e-2, weight_decay=0.1, reduce_bucket_size=700, allgather_bucket_size=700)
    steps(m1, o1, 0, 2)
    sd_opt = o1.state_dict()
    sd_model = {k: v.clone() for k, v in m1.state_dict().items()}
    steps(m1, o1, 2, 4)
    m2 = _model()
    m2.load_state_dict(sd_model)
    o2 = Zero1AdamW(m2.parameters(), lr=1e-2, weight_decay=0.1, reduce_bucket_size=700, allgather_bucket_size=700)
    o2.load_state_dict(sd_opt)
    steps(m2, o2, 2, 4)
    for a, b in zip(m1.parameters(), m2.parameters()):
        assert torch.equal(a, b)
    o3 = Zero1AdamW(_model().parameters(), lr=1e-2)                      # one big bucket: another layout
    with pytest.raises(ValueError):
        o3.load_state_dict(sd_opt)
