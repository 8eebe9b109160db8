"""CPU-only tests of the host side: the C-ABI library loads and exports every symbol include/sfsn.h declares,
weight packing (pure host integer work) is exact to its stated bound, BatchNorm folding is ATen-exact, the drop-in
modules keep the reference's state-dict contract and fail loudly without a GPU, and the clip sharding / all-gather
is correct on a world_size-2 gloo group."""
import os
import re
import socket

import numpy as np
import pytest
import torch

import refweights as rw

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from spiking_fullsubnet_amd import _lib
    _lib.build()
    L = _lib.lib()
    header = open(os.path.join(ROOT, "include", "sfsn.h")).read()
    declared = set(re.findall(r"\b(sfsn_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name
    header_ver = int(re.search(r"#define SFSN_ABI_VERSION (\d+)", header).group(1))
    assert L.sfsn_abi_version() == header_ver == _lib.ABI_VERSION
    L.sfsn_source_hash.restype = __import__('ctypes').c_char_p
    assert L.sfsn_source_hash().decode() == _lib.source_hash()  # the staleness check of _lib.lib()
    assert L.sfsn_strerror(-4).decode().startswith("Number of frequency bins")


def test_no_device_means_loud_failure_not_fallback():
    """Without a GPU every launch returns SFSN_EHIP (-> RuntimeError); nothing silently computes on the CPU."""
    from spiking_fullsubnet_amd import _lib
    import spiking_fullsubnet_amd as pkg
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = _lib.lib()
    assert L.sfsn_device_count() == 0
    m = pkg.SpikingFullSubNet(**rw.LIVE_TINY).eval()
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(1, 4096))
    with pytest.raises(RuntimeError):
        pkg.Engine(m._spec(), {k: v.numpy() for k, v in m.state_dict().items()}, "cpu")
    # argument checks run before any launch: malformed calls are refused on a box without a GPU too
    import ctypes
    buf = ctypes.create_string_buffer(64)
    a = ctypes.addressof(buf)
    assert L.sfsn_cum_laplace_norm(None, 4, 2, 8, None, 0, a, None) == _lib.SFSN_EINVAL
    assert L.sfsn_cum_laplace_norm(a, 4, 2, 8, None, -1, a, None) == _lib.SFSN_EINVAL
    assert L.sfsn_cum_laplace_norm(a, 4, 2, 300, None, 0, a, None) == _lib.SFSN_EUNSUPPORTED  # rows wider than 256 features
    c = pkg.Separator(**rw.FROZEN_TINY_CUM).eval()  # the cumulative-norm front-end constructs (the reference's raises at forward)
    assert c._spec().cum_laplace and not c._spec().laplace


@pytest.mark.parametrize("n,k", [(224, 224), (320, 320), (40, 224), (6, 48), (160, 38), (17, 65)])
def test_w3_pack_roundtrip_bound(n, k):
    """|W - W~| <= dq/2 per row (<= one fp32 ulp of the row's largest binade), dq a power of two, digits in int8,
    zero padding, and exactness for weights already on the 24-bit grid."""
    from spiking_fullsubnet_amd.engine import pack_w3, unpack_w3
    rng = np.random.default_rng(n * 1000 + k)
    w = (rng.standard_normal((n, k)) * rng.uniform(1e-3, 3.0, (n, 1))).astype(np.float32)
    w[0, :] = 0.0
    pk, dq = pack_w3(w)
    assert pk.dtype == np.int8 and pk.size == 3 * ((n + 15) // 16) * ((k + 63) // 64) * 1024
    w2 = unpack_w3(pk, dq, n, k)
    assert (np.abs(w2.astype(np.float64) - w) <= dq[:n, None].astype(np.float64) / 2 + 1e-30).all()
    m, e = np.frexp(dq[:n])
    assert (m == 0.5).all()                                            # powers of two
    assert (np.abs(w).max(1) <= dq[:n].astype(np.float64) * 8355711).all()   # digits cannot overflow
    assert (dq[n:] == 0).all()
    grid = (np.round(w / dq[:n, None]) * dq[:n, None]).astype(np.float32)  # already representable -> exact
    pk2, dq2 = pack_w3(grid)
    np.testing.assert_array_equal(unpack_w3(pk2, dq2, n, k), grid)


def test_w3_pack_rejects_nonfinite():
    from spiking_fullsubnet_amd.engine import pack_w3
    w = np.ones((4, 4), np.float32)
    w[1, 2] = np.inf
    with pytest.raises(ValueError):
        pack_w3(w)


def test_fold_batchnorm_is_aten_exact():
    """alpha/beta reproduce torch's CPU eval BatchNorm1d bit for bit (the oracle restates the same form)."""
    from spiking_fullsubnet_amd.engine import fold_batchnorm
    torch.manual_seed(0)
    C = 224
    bn = torch.nn.BatchNorm1d(C).eval()
    with torch.no_grad():
        bn.running_mean.normal_(0, .5); bn.running_var.uniform_(.3, 1.5); bn.weight.normal_(1, .2); bn.bias.normal_(0, .3)
    x = torch.randn(257, C)
    y = bn(x).detach().numpy()
    a, b = fold_batchnorm(bn.weight.detach().numpy(), bn.bias.detach().numpy(), bn.running_mean.numpy(), bn.running_var.numpy())
    y2 = (x.numpy().astype(np.float64) * a.astype(np.float64) + b.astype(np.float64)).astype(np.float32)  # fma
    np.testing.assert_array_equal(y, y2)


def test_state_dict_contract_live_and_frozen(golden_dir):
    """Same keys / shapes as the reference modules: synthetic reference-named dicts and the zoo checkpoint load strict."""
    import spiking_fullsubnet_amd as pkg
    for kw, seed in ((rw.LIVE_M, 1), (rw.LIVE_TINY_UNSHARED, 2), (rw.LIVE_TINY_2SPK, 3)):
        m = pkg.SpikingFullSubNet(**kw)
        sd = rw.live_state_dict(kw, seed)
        assert set(m.state_dict().keys()) == set(sd.keys())
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    g = np.load(os.path.join(golden_dir, "frozen_s_zoo.npz"))
    s = pkg.Separator(**rw.FROZEN_S)
    zoo = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")}
    assert len(zoo) == 72
    s.load_state_dict(zoo, strict=True)
    assert sum(p.numel() for p in s.parameters()) == 520920  # trainable parameters of baseline_s


def test_constructor_rejections_match_reference():
    import spiking_fullsubnet_amd as pkg
    with pytest.raises(NotImplementedError):
        pkg.SpikingFullSubNet(**dict(rw.LIVE_TINY, sequence_model="GRU"))       # modeling:47
    with pytest.raises(AssertionError):
        pkg.SpikingFullSubNet(**dict(rw.LIVE_TINY, freq_cutoffs=[0, 32, 256]))  # modeling:197
    with pytest.raises(NotImplementedError):
        pkg.Separator(**dict(rw.FROZEN_TINY, sequence_model="LSTM"))            # model_low_freq:70
    with pytest.raises(NotImplementedError):
        pkg.Separator(**dict(rw.FROZEN_TINY, norm_type="forgetting_norm"))      # model_low_freq:227-231


def test_stream_hop_plan_on_the_host():
    """sfsn_hop_stages (host-only part of sfsn_stream_hop): the stage table of a launch for the baseline_m geometry -- stages in
    dependency order, eight wave tiles per workgroup -- and what the launch does not cover."""
    import ctypes
    from spiking_fullsubnet_amd import _lib
    L = _lib.lib()
    kw = rw.LIVE_M
    keep = ctypes.create_string_buffer(64)
    a = ctypes.addressof(keep)  # any non-NULL address: the plan never dereferences device pointers

    def seq(dst, H, P, nl, lo, n_units, ctr, nbr, ctr_fb, nbr_fb, df, fc):
        dst.n_layers, dst.H, dst.P, dst.df, dst.fc = nl, H, P, df, fc
        dst.feat.lo, dst.feat.n_units, dst.feat.ctr, dst.feat.nbr, dst.feat.ctr_fb, dst.feat.nbr_fb = lo, n_units, ctr, nbr, ctr_fb, nbr_fb
        dst.feat.norm, dst.feat.ln_w, dst.feat.ln_b, dst.feat.ln_eps = _lib.NORM_LAYERNORM, a, a, 1e-5
        dst.w_p, dst.w_p_dq, dst.b_p = a, a, a
        for l in range(nl):
            o = dst.layer[l]
            o.w_hh, o.w_hh_dq, o.bias, o.bn_alpha, o.bn_beta, o.c, o.spikes = a, a, a, a, a, a, a
            o.h[0], o.h[1] = a, a
            if l == 0:
                o.w_ih_frag = a
            else:
                o.w_ih, o.w_ih_dq = a, a

    def desc(B=1, hop=1, waveform=False):
        d = _lib.HopDesc()
        seq(d.fb, kw["fb_hidden_size"], kw["fb_proj_size"], kw["fb_num_layers"], 0, 1, kw["fb_input_size"], 0, 0, 0, 0, 0)
        cut, ctr, nbr, df = kw["freq_cutoffs"], kw["center_freq_sizes"], kw["neighbor_freq_sizes"], kw["df_orders"]
        for g in range(3):
            seq(d.sb[g], kw["sb_hidden_size"], 2 * ctr[g] * df[g], kw["sb_num_layers"], cut[g], (cut[g + 1] - cut[g]) // ctr[g], ctr[g], nbr[g],
                ctr[g], 0, df[g], ctr[g])
        d.n_groups, d.B, d.F, d.S, d.hop, d.D, d.fdrc = 3, B, 257, 1, hop, max(df) - 1, 0.5
        d.inp_ri = d.hist_ri = d.enh_ri = d.enh_mag = a
        if waveform:
            d.wave_in = d.wave_state = d.ola_state = d.wave_out = d.window = d.spec_g = d.enh_g = a
        return d

    out = (ctypes.c_int * 128)()
    n = L.sfsn_hop_stages(ctypes.byref(desc()), out, 32)
    stages = [tuple(out[4 * i:4 * i + 4]) for i in range(n)]
    # (sequence, layer, first workgroup, workgroups): 20 / 14 tiles per layer -> 3 / 2 workgroups of eight waves
    assert stages == [(0, 0, 0, 3), (0, 1, 3, 3), (1, 0, 6, 2), (2, 0, 8, 2), (3, 0, 10, 2), (1, 1, 12, 2), (2, 1, 14, 2), (3, 1, 16, 2),
                      (1, -1, 18, 1), (2, -1, 19, 1), (3, -1, 20, 1)]
    n = L.sfsn_hop_stages(ctypes.byref(desc(waveform=True)), out, 32)
    stages = [tuple(out[4 * i:4 * i + 4]) for i in range(n)]
    assert stages[1] == (0, -2, 3, 1) and stages[-1] == (0, -3, 22, 1) and n == 13  # STFT behind the first stage, inverse STFT last
    n = L.sfsn_hop_stages(ctypes.byref(desc(B=3, hop=4)), out, 32)
    assert n == 11 and out[4 * 2 + 3] == 4  # group 0: 24 rows = two row tiles x two workgroups
    assert L.sfsn_hop_stages(ctypes.byref(desc(hop=30)), out, 32) == _lib.SFSN_EUNSUPPORTED          # D + hop > 32
    assert L.sfsn_hop_stages(ctypes.byref(desc(hop=2, waveform=True)), out, 32) == _lib.SFSN_EUNSUPPORTED  # one hop per launch
    bad = desc()
    bad.sb[1].P += 2
    assert L.sfsn_hop_stages(ctypes.byref(bad), out, 32) == _lib.SFSN_EINVAL                         # P != 2 * fc * df * S
    bad = desc()
    bad.fb.layer[1].w_ih = None
    assert L.sfsn_hop_stages(ctypes.byref(bad), out, 32) == _lib.SFSN_EINVAL
    bad = desc()
    bad.fb.feat.norm = _lib.NORM_LAPLACE
    assert L.sfsn_hop_stages(ctypes.byref(bad), out, 32) == _lib.SFSN_EUNSUPPORTED                   # utterance statistics: not causal


def test_training_mode_and_grad_inputs_take_the_differentiable_path_or_raise():
    """Round 3: the live module routes a training-mode call and an input that requires grad to the differentiable path
    (training.py: HIP training-step kernels + ATen) -- which, like everything here, has no CPU path; the frozen competition model
    (its recipe's trainers do not import in the reference either) still refuses both; the inference-only entry points refuse a
    module in training mode."""
    import spiking_fullsubnet_amd as pkg
    y = torch.zeros(1, 2048)
    live = pkg.SpikingFullSubNet(**rw.LIVE_TINY)
    with pytest.raises(RuntimeError, match="no CPU path"):
        live.train()(y)
    with pytest.raises(RuntimeError, match="no CPU path"):
        live.eval()(y.clone().requires_grad_())
    with pytest.raises(RuntimeError, match="no CPU path"):
        live.eval()(y)
    with pytest.raises(RuntimeError, match="training mode"):
        live.train().forward_stft(torch.zeros(1, 257, 4, dtype=torch.complex64))
    frozen = pkg.Separator(**rw.FROZEN_TINY)
    with pytest.raises(RuntimeError, match="no CPU path"):  # (round 4: training mode takes training.forward_frozen -- HIP tensors only)
        frozen.train()(y)
    with pytest.raises(RuntimeError, match="no CPU path"):
        frozen.eval()(y.clone().requires_grad_())
    with pytest.raises(RuntimeError, match="training mode"):
        frozen.train().forward_stft(torch.zeros(1, 257, 4, dtype=torch.complex64))
    with pytest.raises(RuntimeError, match="no CPU path"):
        frozen.eval()(y)


def test_reference_init_is_reproduced():
    """Same RNG consumption order as the reference constructors: torch.manual_seed(s); Model(**kw) gives the same
    initial weights (checked against values recorded from the reference: U(-1/sqrt(H), 1/sqrt(H)) cells first)."""
    import spiking_fullsubnet_amd as pkg
    torch.manual_seed(0)
    m = pkg.SpikingFullSubNet(**rw.LIVE_TINY)
    w = m.fb_model.sequence_model.layers[0].cell.weight_ih
    assert w.shape == (48, 64) and float(w.detach().abs().max()) <= 1 / np.sqrt(48) + 1e-7
    torch.manual_seed(0)
    expect = torch.empty(48, 64).uniform_(-1 / np.sqrt(48), 1 / np.sqrt(48))
    assert torch.equal(w.detach(), expect)  # first RNG draw of the constructor is the first cell's weight_ih


@pytest.mark.parametrize("fname,shared", [("live_tiny.npz", True), ("live_m.npz", True), ("live_tiny_unshared.npz", False),
                                          ("frozen_s_zoo.npz", True), ("frozen_m_zoo.npz", True)])
def test_metric_dropins_match_reference_values(golden_dir, fname, shared):
    """metric.compute_synops / compute_neuronops on the reference's recorded layer outputs == the values the reference's own
    audiozen.metric functions returned for them (recorded by tests/golden/make_golden.py), both from the fp32 spike tensors
    and from SpikeSummary objects (exact counts + shapes, what layer_outputs="counts" returns)."""
    import parity
    from spiking_fullsubnet_amd import SpikeSummary, metric
    gold = np.load(os.path.join(golden_dir, fname))
    if "synops" not in gold:
        pytest.skip("fixture without SynOPs")
    prefixes = ["fb"] + sorted({k.split("/")[0] for k in gold.keys() if k.startswith("sb")}, key=lambda v: int(v[2:]))

    def lists(summary):
        out = []
        for pre in prefixes:
            ent = [torch.from_numpy(gold[f"{pre}/x"])]
            l = 0
            while f"{pre}/spikes_shape/{l}" in gold:
                shape = tuple(int(v) for v in gold[f"{pre}/spikes_shape/{l}"])
                spk = parity.unpack(gold[f"{pre}/spikes_packed/{l}"], shape)
                ent.append(SpikeSummary(torch.tensor(int(spk.sum()), dtype=torch.int64), shape) if summary
                           else torch.from_numpy(spk.astype(np.float32)))
                l += 1
            ent.append(torch.from_numpy(gold[f"{pre}/proj"]))
            out.append(ent)
        return out[0], out[1:]

    for summary in (False, True):
        fb, sb = lists(summary)
        assert metric.compute_synops(fb, sb, shared_weights=shared) == pytest.approx(float(gold["synops"]), rel=1e-6)
        assert metric.compute_neuronops(fb, sb) == float(gold["neuronops"])


def test_checkpoint_bridge_reads_the_reference_formats(golden_dir, tmp_path):
    """accelerate-style checkpoint directories (pytorch_model.bin / model.safetensors, audiozen/trainer.py:225,238-242) load
    strict into the drop-in modules; the trained zoo weights survive the round trip bit for bit; a frozen-named checkpoint
    loads into a shape-compatible live module through the proj <-> fc_output_layer rename."""
    import spiking_fullsubnet_amd as pkg
    from safetensors.torch import save_file
    from spiking_fullsubnet_amd import checkpoint
    g = np.load(os.path.join(golden_dir, "frozen_m_zoo.npz"))
    zoo = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")}
    assert len(zoo) == 72 and sum(v.numel() for k, v in zoo.items() if "num_batches" not in k and "running" not in k) == 953704
    d_bin, d_st = tmp_path / "best", tmp_path / "safe"
    d_bin.mkdir(), d_st.mkdir()
    torch.save(zoo, d_bin / "pytorch_model.bin")
    torch.save({"junk": torch.zeros(1)}, d_bin / "pytorch_model_1.bin")  # the discriminator file must not be picked
    save_file({("module." + k): v.contiguous() for k, v in zoo.items()}, str(d_st / "model.safetensors"))
    for d in (d_bin, d_st, d_bin / "pytorch_model.bin"):
        m = pkg.Separator(**rw.FROZEN_M)
        missing, unexpected = checkpoint.load_checkpoint(m, str(d))
        assert missing == [] and unexpected == []
        for k, v in m.state_dict().items():
            assert torch.equal(v, zoo[k]), k
    live_kw = dict(rw.LIVE_M, use_pre_layer_norm_fb=False, use_pre_layer_norm_sb=False)
    live = pkg.SpikingFullSubNet(**live_kw)
    missing, unexpected = checkpoint.load_checkpoint(live, str(d_bin))
    assert missing == [] and unexpected == []
    assert torch.equal(live.state_dict()["fb_model.proj.weight"], zoo["fb_model.fc_output_layer.weight"])
    with pytest.raises(FileNotFoundError):
        checkpoint.load_checkpoint(live, str(tmp_path / "nope"))
    (tmp_path / "empty").mkdir()
    with pytest.raises(FileNotFoundError):
        checkpoint.find_weights_file(str(tmp_path / "empty"))


def test_shard_bounds_partition():
    from spiking_fullsubnet_amd.dist import shard_bounds
    for n in (0, 1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_clips, q):
    import torch.distributed as dist
    from spiking_fullsubnet_amd.dist import gather_clips, shard_bounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.arange(n_clips * 3 * 5, dtype=torch.float32).reshape(n_clips, 3, 5)
        lo, hi = shard_bounds(n_clips, rank, world)
        out = gather_clips(full[lo:hi].clone(), n_clips)
        q.put((rank, bool(torch.equal(out, full))))
    finally:
        dist.destroy_process_group()


def _run_gather(world, n_clips):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_clips, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(res) == [(r, True) for r in range(world)]


@pytest.mark.parametrize("n_clips", [8, 7])
def test_gather_clips_world2_gloo(n_clips):
    """N > 1 path on CPU: two gloo ranks shard the clips, all-gather and recover the batch in clip order (even and ragged)."""
    _run_gather(2, n_clips)


@pytest.mark.parametrize("n_clips", [512, 21, 5])
def test_gather_clips_world8_gloo(n_clips):
    """BASELINE configs[3]'s shape of the exchange on CPU: EIGHT ranks (one node's worth), 512 clips = 64 per rank (even), 21 clips
    (ragged: five ranks own three, three own two) and fewer clips than ranks (three ranks own none) -- every rank recovers the
    batch in clip order."""
    _run_gather(8, n_clips)


REFERENCE = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference checkout only exists in the build container")
def test_reference_plugin_loader_instantiates_the_dropins_with_bit_identical_init():
    """The drop-in seam: the reference's own loader, `audiozen.utils.instantiate(path, args)` (audiozen/utils.py:113-128,
    called at recipes/intel_ndns/spiking_fullsubnet/run.py:25), builds this package's modules from the recipe's TOML -- only the
    dotted path changes -- and under the same torch.manual_seed the parameters come out BIT-IDENTICAL to the reference module's
    (same names, order, shapes: the checkpoint contract of audiozen/trainer.py:225)."""
    import importlib
    import sys
    import types
    import tomli
    for name in ("librosa", "soundfile", "onnxruntime", "pesq", "pystoi"):  # data / metric deps of audiozen that the path never touches
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__spec__ = importlib.machinery.ModuleSpec(name, None)
            sys.modules[name] = m
    sys.modules["pesq"].pesq = lambda *a, **k: 0.0
    sys.modules["pystoi"].stoi = lambda *a, **k: 0.0
    sys.path.insert(0, REFERENCE)
    try:
        from audiozen.utils import instantiate
        cfg = tomli.load(open(os.path.join(REFERENCE, "recipes/intel_ndns/spiking_fullsubnet/baseline_m.toml"), "rb"))
        args = cfg["model"]["args"]
        assert cfg["model"]["path"] == "audiozen.models.spiking_fullsubnet.modeling_spiking_fullsubnet.SpikingFullSubNet"
        torch.manual_seed(1234)
        ref = instantiate(cfg["model"]["path"], args=args)
        torch.manual_seed(1234)
        mine = instantiate("spiking_fullsubnet_amd.modeling_spiking_fullsubnet.SpikingFullSubNet", args=args)
        import spiking_fullsubnet_amd as pkg
        assert isinstance(mine, pkg.SpikingFullSubNet)
        a, b = ref.state_dict(), mine.state_dict()
        assert list(a.keys()) == list(b.keys())
        for k in a:
            assert a[k].shape == b[k].shape and torch.equal(a[k], b[k]), k
        assert [n for n, _ in ref.named_parameters()] == [n for n, _ in mine.named_parameters()]
        mine.load_state_dict(a, strict=True)  # and a reference checkpoint loads strictly
    finally:
        sys.path.remove(REFERENCE)


def test_training_entry_points_check_their_arguments_without_a_gpu():
    """sfsn_gsn_train_step_* / _seq_*: argument errors come back as codes before anything is launched (runs on a box without a GPU)."""
    import ctypes
    from spiking_fullsubnet_amd import _lib
    L = _lib.lib()
    P, one = ctypes.c_void_p, ctypes.c_void_p(64)  # (a non-null pointer that is never dereferenced on these paths)
    assert L.sfsn_train_scratch_bytes(0) == 0 and L.sfsn_train_scratch_bytes(24) == 0 and L.sfsn_train_scratch_bytes(224) > 0
    # missing tensors
    assert L.sfsn_gsn_train_step_fwd(None, one, one, one, one, None, None, None, None, 0.1, 1e-5, 4, 32, 1, one, one, None, one, one, None, None, 1, None) == _lib.SFSN_EINVAL
    # H not a multiple of the 16-neuron tile
    assert L.sfsn_gsn_train_step_fwd(one, one, one, one, one, None, None, None, None, 0.1, 1e-5, 4, 24, 1, one, one, None, one, one, None, None, 1, None) == _lib.SFSN_EUNSUPPORTED
    # BatchNorm weight without its bias / outputs
    assert L.sfsn_gsn_train_step_fwd(one, one, one, one, one, one, None, None, None, 0.1, 1e-5, 4, 32, 1, one, one, None, one, one, None, None, 1, None) == _lib.SFSN_EINVAL
    # shared gates need d_z; no frames; no zero state
    assert L.sfsn_gsn_train_step_bwd(None, None, one, None, None, one, None, one, one, one, None, None, 4, 32, 1, one, None, one, None, None, None, 1, None) == _lib.SFSN_EINVAL
    assert L.sfsn_gsn_train_seq_fwd(one, one, one, None, None, None, None, 0.1, 1e-5, 0, 4, 32, 1, one, one, one, None, one, one, None, None, None) == _lib.SFSN_EINVAL
    assert L.sfsn_gsn_train_seq_fwd(one, one, one, None, None, None, None, 0.1, 1e-5, 5, 4, 32, 1, None, one, one, None, one, one, None, None, None) == _lib.SFSN_EINVAL
    assert L.sfsn_gsn_train_seq_bwd(one, one, one, None, one, one, None, None, 5, 4, 32, 1, one, one, None, one, None, None, None, None) == _lib.SFSN_EINVAL
    # round 4: the one-launch layer calls (scratch sizing, several calls per launch), the feature launch's zero job, the Gaussian statistics
    assert L.sfsn_train_seq_scratch_bytes(0, 32) == 0 and L.sfsn_train_seq_scratch_bytes(8, 24) == 0
    assert L.sfsn_train_seq_scratch_bytes(512, 224) >= 2 * 512 * 56 * 4 + L.sfsn_train_scratch_bytes(224)
    calls = (_lib.TrainSeqFwd * 2)()
    assert L.sfsn_gsn_train_seq_fwd_multi(calls, 2, 5, 32, 1, None) == _lib.SFSN_EINVAL      # (empty descriptors: missing tensors)
    assert L.sfsn_gsn_train_seq_fwd_multi(calls, 9, 5, 32, 1, None) == _lib.SFSN_EINVAL      # more calls than a launch holds
    assert L.sfsn_gsn_train_seq_bwd_multi((_lib.TrainSeqBwd * 1)(), 1, 0, 32, 1, None) == _lib.SFSN_EINVAL
    Rs = (ctypes.c_int * 2)(8, 0)
    assert L.sfsn_gsn_train_multi_check(Rs, 2, 32, 1) == _lib.SFSN_EINVAL and L.sfsn_gsn_train_multi_check(Rs, 1, 24, 1) == _lib.SFSN_EUNSUPPORTED
    g = (_lib.FeatureGroup * 1)()
    assert L.sfsn_features_z(one, None, 1, 33, 8, 0, 0.5, g, 1, 0, 8, ctypes.c_void_p(64), 24, None) == _lib.SFSN_EINVAL    # not whole 16-byte pieces
    assert L.sfsn_features_z(one, None, 1, 33, 8, 0, 0.5, g, 1, 0, 8, None, 16, None) == _lib.SFSN_EINVAL                   # bytes without a buffer
    assert L.sfsn_gaussian_stats(one, None, 1, 33, 1, 0, 0.5, g, 1, one, one, one, None) == _lib.SFSN_EINVAL                # one frame: no unbiased estimate
    assert L.sfsn_gaussian_stats(one, None, 1, 33, 8, 0, 0.5, g, 1, one, None, one, None) == _lib.SFSN_EINVAL
    # round 5: chunked layer calls (h0 and c0 come together), the feature + input-product launch
    c = (_lib.TrainSeqFwd * 1)()
    for k in ("z", "w_hh", "bias", "spikes", "u", "f", "g", "scratch"):
        setattr(c[0], k, 64)
    c[0].R, c[0].h0 = 8, 64
    assert L.sfsn_gsn_train_seq_fwd_multi(c, 1, 5, 32, 1, None) == _lib.SFSN_EINVAL          # h0 without c0
    assert L.sfsn_scan_split_scratch_bytes(0, 320) == 0 and L.sfsn_scan_split_scratch_bytes(64, 320) == 64 + 2 * 64 * 80 * 4
    sg = (_lib.ScanSegment * 1)()
    assert L.sfsn_gsn_layer_scan_split(sg, 1, 5, 320, 0, None, 0, None) == _lib.SFSN_EINVAL        # no scratch
    assert L.sfsn_gsn_layer_scan_split(sg, 1, 5, 224, 0, one, 1 << 20, None) == _lib.SFSN_EUNSUPPORTED  # one compute unit serves it
    j = (_lib.FeatProjJob * 1)()
    j[0].feat.n_units, j[0].feat.ctr = 1, 32
    assert L.sfsn_features_proj(one, None, 1, 33, 8, 0, 0.5, j, 1, 0, 8, None, 0, None) == _lib.SFSN_EINVAL      # a job that produces nothing
    assert L.sfsn_features_proj(one, None, 1, 33, 8, 0, 0.5, j, 1, 4, 8, None, 0, None) == _lib.SFSN_EINVAL      # frames past the end
    assert L.sfsn_features_proj(one, None, 1, 33, 8, 0, 0.5, j, 0, 0, 8, None, 0, None) == _lib.SFSN_EINVAL      # no jobs
    j[0].feat.x, j[0].w, j[0].z, j[0].H, j[0].ldz = 64, 64, 64, 30, 30
    assert L.sfsn_features_proj(one, None, 1, 33, 8, 0, 0.5, j, 1, 0, 8, None, 0, None) == _lib.SFSN_EUNSUPPORTED  # H % 4 (and 8 rows: not the bf16-split kernel's shape)


@pytest.mark.parametrize("n_fft,hop,wl,T,length", [(512, 128, 512, 40, 128 * 39), (512, 128, 512, 40, None), (256, 64, 256, 17, 64 * 16 - 5),
                                                   (512, 128, 400, 20, 128 * 19), (512, 256, 512, 9, 256 * 8 + 100), (512, 160, 512, 12, 160 * 11)])
def test_training_istft_without_the_host_check_equals_torch_istft(n_fft, hop, wl, T, length):
    """training._istft (no device-to-host read: capturable in a HIP graph) against torch.istft, values and gradient."""
    import torch
    from spiking_fullsubnet_amd.training import _istft
    torch.manual_seed(0)
    w = torch.hann_window(wl)
    x = torch.randn(3, n_fft // 2 + 1, T, dtype=torch.complex64).requires_grad_()
    a = torch.istft(x, n_fft, hop, wl, window=w, length=length)
    b = _istft(x, n_fft, hop, wl, w, length)
    assert a.shape == b.shape
    assert float((a - b).abs().max()) <= 2e-7 * float(a.abs().max()) + 1e-9
    ga, = torch.autograd.grad(a.pow(2).sum(), x)
    gb, = torch.autograd.grad(b.pow(2).sum(), x)
    assert float((ga - gb).abs().max()) <= 1e-6 * float(ga.abs().max())
