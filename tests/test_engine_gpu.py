"""End-to-end parity of the HIP engine (through the C ABI) against the CPU oracle.

Tolerance (BASELINE.json north_star): fp32 rtol 1e-3 on logits; written here as
|hip - oracle| <= 1e-3*|oracle| + 1e-5*max|oracle| (the absolute term only keeps logits that happen to sit near zero
from demanding more than fp32 has; measured error: 3e-7 of the scale in f32, 5e-6 in bf16x3).
"""
import json

import numpy as np
import pytest
import torch

from oracle import tsm_oracle
from tests._util import assert_close, make_input

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def engine224(hip_lib, sd0):
    from workoutdetector_amd.engine import TsmEngine
    eng = TsmEngine(num_class=12, num_segments=8, height=224, width=224, max_clips=4, state_dict=sd0)
    yield eng
    eng.close()


def test_logits_parity_224(engine224, sd0):
    x = make_input(100, 2, 8, 224, 224)
    want = tsm_oracle.tsm_forward(sd0, torch.from_numpy(x)).numpy()
    got = engine224.run(None, {engine224.get_inputs()[0].name: x})[0]
    assert got.shape == (2, 12) and got.dtype == np.float32
    assert_close(got, want, rtol=1e-3, atol_scale=1e-5, what='logits 224')


def test_stage_taps_224(engine224, sd0):
    """Every stage of TSM.forward, so a wrong sub-stage cannot hide behind the final tolerance."""
    x = make_input(7, 1, 8, 224, 224)
    taps = {}
    tsm_oracle.tsm_forward(sd0, torch.from_numpy(x), taps=taps)
    for stage in ['stem', 'layer1.0', 'layer1.2', 'layer2.0', 'layer2.3', 'layer3.0', 'layer3.5', 'layer4.0',
                  'layer4.2']:
        got = engine224.forward_tap(x, stage)
        want = taps[stage].permute(0, 2, 3, 1).numpy()
        assert_close(got, want, rtol=1e-3, atol_scale=1e-5, what=stage)


def test_golden_logits(hip_lib, golden_dir):
    """Committed oracle logits (tests/golden/tsm_r50_logits.json): all shapes incl. T=16 and non-square."""
    from workoutdetector_amd.engine import TsmEngine
    from workoutdetector_amd.weights import make_state_dict
    gold = json.load(open(f'{golden_dir}/tsm_r50_logits.json'))
    for name, case in gold.items():
        b, t, _, h, w = case['shape']
        eng = TsmEngine(num_class=12, num_segments=t, height=h, width=w, max_clips=b,
                        state_dict=make_state_dict(case['weight_seed'], 12))
        x = make_input(case['input_seed'], b, t, h, w)
        got = eng.run(None, {'input': x})[0]
        assert_close(got, np.array(case['logits'], dtype=np.float32), rtol=1e-3, atol_scale=1e-5, what=name)
        eng.close()


def test_module_duck_type_and_device_path(engine224, sd0):
    """nn.Module style call with [B*T,3,H,W]; host, device and NTHWC inputs agree bit-for-bit."""
    from workoutdetector_amd import _lib
    x = make_input(3, 3, 8, 224, 224)
    host = engine224(torch.from_numpy(x).reshape(24, 3, 224, 224))
    assert isinstance(host, torch.Tensor) and tuple(host.shape) == (3, 12)
    dev = engine224(torch.from_numpy(x).cuda().reshape(24, 3, 224, 224))
    assert dev.is_cuda
    assert torch.equal(dev.cpu(), host)
    nthwc = np.ascontiguousarray(x.transpose(0, 1, 3, 4, 2))
    assert np.array_equal(engine224.forward_host(nthwc, layout=_lib.LAYOUT_NTHWC), host.numpy())


def test_batch_chunking_and_independence(engine224):
    """B > max_clips is chunked; a clip's logits do not depend on its batch neighbours (the temporal
    shift never crosses a clip boundary)."""
    x = make_input(11, 6, 8, 224, 224)
    full = engine224.run(None, {'input': x})[0]
    for i in (0, 3, 5):
        single = engine224.run(None, {'input': x[i:i + 1]})[0]
        assert np.array_equal(single[0], full[i])


def test_no_shift_engine(hip_lib, sd0):
    from workoutdetector_amd.engine import TsmEngine
    eng = TsmEngine(num_class=12, num_segments=8, height=64, width=64, max_clips=2, is_shift=False, state_dict=sd0)
    x = make_input(5, 2, 8, 64, 64)
    want = tsm_oracle.tsm_forward(sd0, torch.from_numpy(x), is_shift=False).numpy()
    assert_close(eng.run(None, {'input': x})[0], want, rtol=1e-3, atol_scale=1e-5, what='no-shift')
    eng.close()


def test_error_behaviour(hip_lib, sd0, engine224):
    from workoutdetector_amd._lib import TsmError
    from workoutdetector_amd.engine import TsmEngine
    with pytest.raises(ValueError):
        engine224.run(None, {'input': np.zeros((1, 8, 3, 200, 224), np.float32)})
    with pytest.raises(ValueError):
        engine224.run(None, {'wrong': np.zeros((1, 8, 3, 224, 224), np.float32)})
    # the C ABI takes bare pointers: buffer sizes are checked on the Python side before anything is launched
    from workoutdetector_amd import _lib
    with pytest.raises(ValueError):
        engine224.forward_device(torch.zeros(1, 8, 3, 224, 200, device='cuda'))
    with pytest.raises(ValueError):
        engine224.forward_device(torch.zeros(1, 8, 224, 224, 3, device='cuda'), layout=_lib.LAYOUT_NTHWC4)
    with pytest.raises(ValueError):
        engine224.forward_device(torch.zeros(1, 8, 3, 224, 224, device='cuda'), out=torch.zeros(2, 12, device='cuda'))
    with pytest.raises(ValueError):
        engine224.forward_host(np.zeros((2, 8, 3, 224, 112), np.float32))
    eng = TsmEngine(max_clips=1)
    with pytest.raises(TsmError) as ei:
        eng.forward_host(np.zeros((1, 8, 3, 224, 224), np.float32))
    assert ei.value.status == -3
    bad = dict(sd0)
    del bad['base_model.layer3.2.conv2.weight']
    with pytest.raises(TsmError) as ei:
        eng.load_state_dict(bad)
    assert ei.value.status == -4 and 'layer3.2.conv2' in str(ei.value)
    eng.close()
    eng = TsmEngine(max_clips=1)
    bad = dict(sd0)
    bad['fc.weight'] = torch.zeros(5, 2048)
    with pytest.raises(TsmError) as ei:
        eng.load_state_dict(bad)
    assert ei.value.status == -5
    eng.close()


def test_autotuned_tiles_are_bitwise_invariant(hip_lib, sd0, monkeypatch):
    """The engine picks a conv tile shape per layer by timing; every shape accumulates each output in the
    same k order, so logits must not depend on the choice (heuristic vs tuned vs each forced shape)."""
    from workoutdetector_amd.engine import TsmEngine, launch_trace
    from tests._util import IGEMM_TILE_DIMS, assert_ran_tile
    x = make_input(21, 3, 8, 96, 96)
    outs = {}
    for name, env in [('tuned', {}), ('heuristic', {'TSM_AUTOTUNE': '0'}),
                      ('64x64', {'TSM_AUTOTUNE': '0', 'TSM_CONV_TILE': '64x64'}),
                      ('32x32', {'TSM_AUTOTUNE': '0', 'TSM_CONV_TILE': '32x32'}),
                      ('128x128w8', {'TSM_AUTOTUNE': '0', 'TSM_CONV_TILE': '128x128w8'})]:
        for k in ('TSM_AUTOTUNE', 'TSM_CONV_TILE'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = TsmEngine(height=96, width=96, max_clips=3, state_dict=sd0)
        outs[name] = eng.run(None, {'input': x})[0]
        with launch_trace() as tr:
            again = eng.run(None, {'input': x})[0]          # second call runs from the tile cache
        assert np.array_equal(outs[name], again)
        eng.close()
        if name in IGEMM_TILE_DIMS:       # the forced shape ran (wherever it is valid; the rest keeps the heuristic shape)
            assert_ran_tile(tr, name, f'f32 engine forced onto {name}')
    assert np.array_equal(outs['tuned'], outs['heuristic'])
    assert np.array_equal(outs['tuned'], outs['64x64']) and np.array_equal(outs['tuned'], outs['32x32'])
    assert np.array_equal(outs['tuned'], outs['128x128w8'])


def test_create_model_from_checkpoint_and_onnx(hip_lib, sd0, tmp_path):
    """The four weight sources give the same logits: state dict, torch checkpoint with 'module.' keys
    (create_model's remap, tsm.py:451-473), an mmaction2-style checkpoint and the reference's deployment artefact: an
    eval-mode ``torch.onnx.export`` (opset 11, BatchNorm fused) written by torch's exporter (tests/_torch_tsm.py)."""
    from tests._torch_tsm import LitWrapper, TorchTSM, export_onnx
    from workoutdetector_amd.engine import TsmEngine, create_model
    x = make_input(9, 1, 8, 64, 64)
    ref = TsmEngine(height=64, width=64, max_clips=1, state_dict=sd0)
    want = ref.run(None, {'input': x})[0]
    ref.close()
    # the oracle leg: every imported weight source is also held against the CPU oracle on the ORIGINAL state dict
    # (an import that permuted or dropped a tensor identically for all four sources would still agree engine-to-engine)
    oracle_want = tsm_oracle.tsm_forward(sd0, torch.from_numpy(x)).numpy()
    assert_close(want, oracle_want, rtol=1e-3, atol_scale=1e-5, what='state dict vs oracle')
    keys = list(sd0)
    ck = {'state_dict': {('module.' + k).replace('module.fc.', 'module.new_fc.'): sd0[k] for k in keys}}
    torch.save(ck, tmp_path / 'tsm.pth')
    m = create_model(num_class=12, checkpoint=str(tmp_path / 'tsm.pth'), device='cuda:0', height=64, width=64, max_clips=1)
    got = m.run(None, {'input': x})[0]
    assert np.array_equal(got, want)
    assert_close(got, oracle_want, rtol=1e-3, atol_scale=1e-5, what='.pth checkpoint vs oracle')
    m.close()
    from tests.test_weights import _to_mmaction          # mmaction2 checkpoint of the reference's --mmlab branch
    torch.save({'meta': {}, 'state_dict': _to_mmaction(sd0)}, tmp_path / 'tsm_mmaction.pth')
    m = create_model(num_class=12, checkpoint=str(tmp_path / 'tsm_mmaction.pth'), device='cuda:0', height=64, width=64,
                     max_clips=1)
    got = m.run(None, {'input': x})[0]
    assert np.array_equal(got, want)
    assert_close(got, oracle_want, rtol=1e-3, atol_scale=1e-5, what='mmaction checkpoint vs oracle')
    m.close()
    export_onnx(LitWrapper(TorchTSM(num_class=12).load_engine_state_dict(sd0)), str(tmp_path / 'tsm.onnx'),
                sample_shape=(1, 8, 3, 64, 64))
    m = create_model(num_class=12, checkpoint=str(tmp_path / 'tsm.onnx'), device='cuda:0', height=64, width=64, max_clips=1)
    got = m.run(None, {'input': x})[0]
    assert_close(got, want, rtol=1e-5, atol_scale=1e-6, what='onnx-imported weights')
    assert_close(got, oracle_want, rtol=1e-3, atol_scale=1e-5, what='onnx-imported weights vs oracle')
    m.close()


def test_c_program_runs_a_forward(hip_lib, tmp_path):
    """The engine driven from plain C (tests/abi_c_smoke.c): all-zero convs -> logits == fc.bias exactly."""
    import subprocess
    from tests.test_abi import _build_c_smoke
    exe = _build_c_smoke(tmp_path)
    out = subprocess.run([exe, 'run'], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert 'forward ok' in out.stdout


def test_c_program_runs_the_device_pipeline(hip_lib, tmp_path):
    """the dataset loop's device pipeline driven from plain C99 -- tsm_preprocess -> tsm_gather_clips ->
    tsm_forward(TSM_LAYOUT_NTHWC4, device memory) -> tsm_scores_to_states (tests/abi_c_smoke.c `pipeline`) -- against the
    Python binding on the same weights and the same uint8 video: logits, states and top scores bit for bit."""
    import struct
    import subprocess
    import torch
    from tests.test_abi import _build_c_smoke
    from workoutdetector_amd import _lib, engine
    from workoutdetector_amd.weights import make_state_dict
    sd = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in make_state_dict(3, 12).items()
          if not k.endswith('num_batches_tracked')}
    sd['fc.weight'] = sd['fc.weight'] * 40.0          # spread the softmax so that states are not all -1
    frames, h, w, resize, crop = 43, 70, 90, 72, 64   # 6 clips, the last one zero-padded; non-square source
    video = np.random.default_rng(9).integers(0, 256, (frames, h, w, 3), dtype=np.uint8)
    with open(tmp_path / 'w.bin', 'wb') as f:
        f.write(struct.pack('<i', len(sd)))
        for k, v in sd.items():
            f.write(struct.pack('<i', len(k)) + k.encode() + struct.pack('<i', v.ndim) + struct.pack(f'<{v.ndim}q', *v.shape))
            f.write(v.tobytes())
    with open(tmp_path / 'v.bin', 'wb') as f:
        f.write(struct.pack('<5i', frames, h, w, resize, crop) + video.tobytes())
    exe = _build_c_smoke(tmp_path)
    out = subprocess.run([exe, 'pipeline', str(tmp_path / 'w.bin'), str(tmp_path / 'v.bin'), str(tmp_path / 'o.bin')],
                         capture_output=True, text=True)
    assert out.returncode == 0 and 'pipeline ok' in out.stdout, out.stdout + out.stderr
    raw = open(tmp_path / 'o.bin', 'rb').read()
    n_clips = struct.unpack_from('<i', raw)[0]
    assert n_clips == (frames + 7) // 8
    c_logits = np.frombuffer(raw, np.float32, n_clips * 12, 4).reshape(n_clips, 12)
    c_states = np.frombuffer(raw, np.int32, n_clips, 4 + n_clips * 48)
    c_top = np.frombuffer(raw, np.float32, n_clips, 4 + n_clips * 52)
    # the same steps through the Python binding
    eng = engine.TsmEngine(num_class=12, num_segments=8, height=crop, width=crop, max_clips=n_clips, state_dict=sd)
    even = torch.cat([torch.from_numpy(video[0::2]), torch.zeros((1, h, w, 3), dtype=torch.uint8)]).cuda()
    packed = engine.preprocess_frames(even, resize=resize, crop=crop, layout=_lib.LAYOUT_NTHWC4)
    clips = engine.gather_clips(packed, 0, frames, 0, n_clips)
    logits = eng.forward_device(clips, layout=_lib.LAYOUT_NTHWC4)
    states, top = engine.scores_to_states(logits, threshold=0.1, return_top=True)
    assert np.array_equal(c_logits, logits.cpu().numpy())
    assert np.array_equal(c_states, states.cpu().numpy()) and np.array_equal(c_top, top.cpu().numpy())
    assert (c_states >= 0).any(), c_states
    eng.close()


@pytest.mark.parametrize('dtype,t,div,h,w,ncls,rtol', [
    ('f32', 4, 16, 96, 128, 5, 1e-3),        # fold = C/16, non-square, odd class count
    ('f32', 3, 8, 64, 96, 2, 1e-3),          # odd segment count (the reference's factory default num_class=2)
    ('f32', 1, 8, 64, 64, 12, 1e-3),         # single-frame clips: both shifted channel groups read zeros
    ('bf16x3', 4, 8, 96, 64, 7, 1e-3),
    ('bf16', 2, 8, 64, 64, 12, 1e-2),
    ('f32', 2, 8, 270, 480, 12, 1e-3),       # wide frames: many pooled tiles per row, ragged in both directions
    ('bf16x3', 2, 8, 270, 480, 12, 1e-3),
    # the round-2 kernels forced on (the tuner would otherwise decide by timing): fused conv2 + conv3, 256 x 256 LDS-DMA tile
    ('f32+fused', 3, 8, 64, 96, 2, 1e-3),
    ('f32+fused', 4, 16, 96, 128, 5, 1e-3),
    ('bf16x3+fused', 4, 8, 96, 64, 7, 1e-3),
    ('bf16+fused', 4, 8, 96, 64, 7, 1e-2),
    ('bf16+fused', 3, 8, 90, 70, 12, 1e-2),
    ('bf16+256x256', 2, 8, 64, 64, 12, 1e-2),
    ('bf16+256x256', 3, 8, 90, 70, 12, 1e-2),
    ('bf16+256x256p', 2, 8, 64, 64, 12, 1e-2),   # round 3: the persistent form of that tile
    ('bf16+256x256p', 3, 8, 90, 70, 12, 1e-2),
])
def test_unusual_configurations_against_oracle(hip_lib, monkeypatch, capsys, dtype, t, div, h, w, ncls, rtol):
    """Segment counts, shift_div, class counts and aspect ratios other than the headline's, vs the CPU oracle."""
    from workoutdetector_amd.engine import TsmEngine, launch_trace
    from workoutdetector_amd.weights import make_state_dict, to_torch
    dtype, _, force = dtype.partition('+')
    if force == 'fused':
        monkeypatch.setenv('TSM_FUSE_CONV23', '1')
        monkeypatch.setenv('TSM_FUSE_BLOCK', '0')     # (bf16: the whole-block kernel would otherwise take layer1 away from it)
    elif force:
        monkeypatch.setenv('TSM_AUTOTUNE', '0')
        monkeypatch.setenv('TSM_CONV_TILE', force)
    sd = make_state_dict(7, ncls)
    eng = TsmEngine(num_class=ncls, num_segments=t, height=h, width=w, shift_div=div, max_clips=3, state_dict=sd,
                    dtype=dtype)
    x = make_input(100 + t, 3, t, h, w)
    with launch_trace() as tr:
        got = eng.run(None, {'input': x})[0]
    eng.close()
    if force == 'fused' and dtype == 'bf16':     # conv2 + conv3 as one launch: the weight-stationary form (layer1.1-2)
        assert tr.count('conv3x3_ws_kernel<true>') >= 2, sorted(set(tr.kernels))
    elif force == 'fused':                       # ... conv23_fused_kernel (fp32 / split-bf16: layer1.1-2 and layer2.1-3)
        assert tr.ran('conv23_fused_kernel<64,') and tr.ran('conv23_fused_kernel<128,'), sorted(set(tr.kernels))
    elif force:
        from tests._util import assert_ran_tile
        assert_ran_tile(tr, force, f'{dtype} engine forced onto {force}')
    want = tsm_oracle.tsm_forward(to_torch(sd), torch.from_numpy(x), n_segment=t, shift_div=div).numpy()
    assert got.shape == (3, ncls)
    if dtype == 'bf16':      # its own oracle: the bf16-storage restatement (bar = 1e-2 of the logit scale, same arg-max)
        from tests._util import BF16_E2E_BAR, bf16_logits_report
        assert rtol == BF16_E2E_BAR
        want_bf16 = tsm_oracle.tsm_forward_bf16(to_torch(sd), torch.from_numpy(x), n_segment=t, shift_div=div).numpy()
        bf16_logits_report(got, want_bf16, want, f'bf16 T={t} div={div} {h}x{w}', capsys)
        return
    assert_close(got, want, rtol=rtol, atol_scale=1e-5, what=f'{dtype} T={t} div={div} {h}x{w}')


def test_split_k_is_bitwise_identical_to_whole_k(hip_lib, sd0, monkeypatch):
    """Long-K fp32 layers accumulate K in fixed segments (ConvParams::kseg_len), so the split-K launch form (one
    workgroup per tile and segment + ordered reduction, what the tuner picks at small batch) must reproduce the
    whole-K form bit for bit, on both tile shapes that implement it."""
    from workoutdetector_amd.engine import TsmEngine, launch_trace
    x = make_input(31, 2, 8, 96, 96)
    outs = {}
    for name, env in [('whole 64x64', {'TSM_AUTOTUNE': '1', 'TSM_CONV_CODE': '3'}),
                      ('split 64x64', {'TSM_AUTOTUNE': '1', 'TSM_CONV_CODE': str(0x100 | 3)}),
                      ('split 32x32', {'TSM_AUTOTUNE': '1', 'TSM_CONV_CODE': str(0x100 | 4)}),
                      ('tuned', {})]:
        for k in ('TSM_AUTOTUNE', 'TSM_CONV_CODE'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = TsmEngine(height=96, width=96, max_clips=2, state_dict=sd0)
        outs[name] = eng.run(None, {'input': x})[0]      # (the first forward of a bucket also runs the tuning pass, which times BOTH forms)
        with launch_trace() as tr:
            assert np.array_equal(eng.run(None, {'input': x})[0], outs[name])
        # the forced form ran: one splitk_reduce launch behind every segmented layer (tile shape as forced), none in the whole-K form
        if name.startswith('split'):
            assert tr.ran('splitk_reduce_kernel'), sorted(set(tr.kernels))
            dims = '[BM = 64, BN = 64,' if '64x64' in name else '[BM = 32, BN = 32,'
            assert any(k.startswith('conv_igemm<') and 'true>' in k.split('[')[0] and dims in k for k in tr.kernels), sorted(set(tr.kernels))
        elif name.startswith('whole'):
            assert not tr.ran('splitk_reduce_kernel'), sorted(set(tr.kernels))
        tap = eng.forward_tap(x, 'layer4.1')
        outs[name + ' tap'] = tap
        if name == 'tuned':     # 2 clips of 96x96: layer3/4 have a handful of tiles, so split-K normally wins the timing
            picked = sorted(k for k, v in eng.conv_tiles(2).items() if v.endswith('/splitK'))
            print(f'\n[split-K picked by the tuner for] {picked}')      # informational: the choice is timing-based
        eng.close()
    for name in outs:
        ref = outs['whole 64x64 tap' if name.endswith(' tap') else 'whole 64x64']
        assert np.array_equal(ref, outs[name]), name


@pytest.mark.parametrize('clips', [14, 27])
def test_tail_split_is_bitwise_identical_to_whole_k(hip_lib, sd0, monkeypatch, clips):
    """ConvParams::ksplit = 2 (tile code bit 0x200): the tiles of the last, partly filled round of resident workgroups of a
    segmented fp32 launch run as (tile, K segment) pieces + the ordered reduction over the tail rows only.  14 clips of
    224x224 put 1372 / 2744 tiles of 64x64 on the long-K layers of layer2-4 (rounds of 5 x 256 workgroups: the tail is 7-14 %
    of a round) and 688 on layer4's 3x3 (no whole round: the switch must fall back to the whole-K form there)."""
    from workoutdetector_amd.engine import TsmEngine, launch_trace
    # (27 clips: layer3's long-K launches are 2 646 tiles = 2.07 rounds -- a 7 % tail behind two whole rounds --, layer4's 1 323 = 1.03)
    x = make_input(41, clips, 8, 224, 224)
    outs = {}
    for name, code in [('whole', 3), ('tail', 0x200 | 3)]:
        monkeypatch.setenv('TSM_AUTOTUNE', '1')
        monkeypatch.setenv('TSM_CONV_CODE', str(code))
        eng = TsmEngine(height=224, width=224, max_clips=clips, state_dict=sd0)
        outs[name] = eng.run(None, {'input': x})[0]
        with launch_trace() as tr:
            assert np.array_equal(eng.run(None, {'input': x})[0], outs[name])
        n_reduce = tr.count('splitk_reduce_kernel')
        if name == 'tail':
            if torch.cuda.get_device_properties(0).multi_processor_count == 256:
                # layer3: 5 x conv1 (K = 1024) + 6 x conv2, layer4.0.conv1, layer4.0's conv3 + downsample GEMM, ... -- and NOT layer4's
                # 3x3 / conv1 launches at 7x7 (688 tiles: no whole round), which keep the whole-K form
                assert 10 <= n_reduce <= 22, (n_reduce, sorted(set(tr.kernels)))
            else:
                assert n_reduce >= 1, sorted(set(tr.kernels))
        else:
            assert n_reduce == 0
        outs[name + ' tap'] = eng.forward_tap(x, 'layer3.2')
        eng.close()
    assert np.array_equal(outs['whole'], outs['tail'])
    assert np.array_equal(outs['whole tap'], outs['tail tap'])


def test_warmup_tunes_every_bucket(hip_lib, sd0):
    from workoutdetector_amd.engine import TsmEngine
    eng = TsmEngine(height=64, width=64, max_clips=5, state_dict=sd0).warmup()
    for n in (1, 2, 4, 5):
        tiles = eng.conv_tiles(n)
        # (the stem runs on its own pool-fused kernel and the downsample convs inside the fused conv3: neither is tuned)
        assert tiles and all(v != 'heuristic' for k, v in tiles.items() if 'downsample' not in k and k != 'conv1'), (n, tiles)
    eng.close()


def test_cached_split_choice_is_rechecked_against_the_scratch_size(hip_lib, sd0):
    """Tile choices are cached per power-of-two bucket of the clip count: a split-K choice tuned on 5 clips is reused
    for 8, where the segment sums of some layer no longer fit the scratch buffer -> that launch must fall back to the
    whole-K form (same bits), not overrun the buffer."""
    from workoutdetector_amd.engine import TsmEngine
    x = make_input(41, 8, 8, 224, 224)
    eng = TsmEngine(max_clips=8, state_dict=sd0)
    five = eng.run(None, {'input': x[:5]})[0]          # tunes bucket 8 on 5 clips
    eight = eng.run(None, {'input': x})[0]
    assert np.array_equal(five, eight[:5])
    one = eng.run(None, {'input': x[7:8]})[0]
    assert np.array_equal(one[0], eight[7])
    eng.close()


def test_tune_cache_file_round_trip(hip_lib, sd0, tmp_path, monkeypatch):
    """TSM_TUNE_CACHE: the first engine tunes and appends one line per bucket; a second engine (a later process in
    practice) reads the codes instead of timing again; a corrupt file is ignored; results are identical either way."""
    from workoutdetector_amd.engine import TsmEngine
    path = tmp_path / 'tiles.txt'
    monkeypatch.setenv('TSM_TUNE_CACHE', str(path))
    x = make_input(51, 3, 8, 96, 96)
    a = TsmEngine(height=96, width=96, max_clips=4, state_dict=sd0)
    ya = a.run(None, {'input': x})[0]
    tiles_a = a.conv_tiles(3)
    a.close()
    lines = path.read_text().splitlines()
    assert len(lines) == 1 and ' 96x96 ' in lines[0] and lines[0].count('|') == 2
    b = TsmEngine(height=96, width=96, max_clips=4, state_dict=sd0)
    yb = b.run(None, {'input': x})[0]
    assert b.conv_tiles(3) == tiles_a and np.array_equal(ya, yb)
    b.run(None, {'input': x[:1]})                              # another bucket: tuned and appended
    b.close()
    assert len(path.read_text().splitlines()) == 2
    path.write_text(lines[0].rsplit('|', 1)[0] + '|9999,abc\n')     # same key, malformed codes
    c = TsmEngine(height=96, width=96, max_clips=4, state_dict=sd0)
    assert np.array_equal(c.run(None, {'input': x})[0], ya)          # ignored -> tuned again -> appended
    c.close()
    assert len(path.read_text().splitlines()) == 2


def test_tune_cache_keeps_the_fusion_bits(hip_lib, sd0, tmp_path, monkeypatch):
    """A cached line whose codes carry the fusion bits (+1024: conv2 + conv3 as one launch, +2048: the whole block as one
    launch, +4096: conv3 also runs the next block's conv1) must be READ BACK, not rejected: the ranks of a multi-GPU job share one tune cache (rank 0 tunes, the others
    read), and a parser that dropped such lines would make every rank tune for itself.  A bf16 engine at a batch where the
    tuner picks the whole-block kernel writes its line; the line is then edited (every layer1 / layer2 fusion bit the
    layer supports is set or cleared by hand) and a second engine must report exactly the edited codes."""
    from workoutdetector_amd.engine import TsmEngine
    path = tmp_path / 'tiles.txt'
    monkeypatch.setenv('TSM_TUNE_CACHE', str(path))
    x = make_input(52, 32, 8, 224, 224)[:32]
    a = TsmEngine(max_clips=32, state_dict=sd0, dtype='bf16')
    ya = a.run(None, {'input': x})[0]
    tiles_a = a.conv_tiles(32)
    a.close()
    line = path.read_text().splitlines()[0]
    key, codes = line.rsplit('|', 1)
    codes = [int(c) for c in codes.split(',')]
    assert any(c & 0x800 for c in codes) or any(c & 0x400 for c in codes), tiles_a     # the tuner uses a fused form at this size
    # the line is in the engine's layer order: stem, then per block conv1, conv2, conv3 (, downsample)
    order = ['conv1'] + [f'layer{li}.{b}.{part}' for li, nb in enumerate((3, 4, 6, 3), 1) for b in range(nb)
                         for part in ('conv1', 'conv2', 'conv3') + (('downsample',) if b == 0 else ())]
    assert len(order) == len(codes) == 53
    flipped = list(codes)
    for i, nme in enumerate(order):                      # toggle the whole-block bit of the three layer1 blocks
        if nme in ('layer1.0.conv1', 'layer1.1.conv1', 'layer1.2.conv1'):
            flipped[i] ^= 0x800
    path.write_text(key + '|' + ','.join(str(c) for c in flipped) + '\n')
    b = TsmEngine(max_clips=32, state_dict=sd0, dtype='bf16')
    yb = b.run(None, {'input': x})[0]
    tiles_b = b.conv_tiles(32)
    b.close()
    assert len(path.read_text().splitlines()) == 1, 'the edited line was rejected and the engine tuned again'
    for nme in ('layer1.0.conv1', 'layer1.1.conv1', 'layer1.2.conv1'):
        assert tiles_b[nme].endswith('+block') != tiles_a[nme].endswith('+block'), (nme, tiles_a[nme], tiles_b[nme])
    assert np.array_equal(ya, yb)                         # and, as always, not a bit of the result depends on it


@pytest.mark.parametrize('dtype', ['f32', 'bf16x3', 'bf16'])
@pytest.mark.parametrize('h,w,b', [(224, 224, 2), (96, 96, 3), (90, 70, 1), (256, 256, 3)])
def test_fused_conv2_conv3_equals_the_separate_kernels_bitwise(hip_lib, sd0, monkeypatch, h, w, b, dtype):
    """conv23_fused (Bottleneck.conv2 + bn2 + ReLU + conv3 + bn3 + residual + ReLU in one kernel, the mid tensor kept in
    LDS; layer1.1-2 and layer2.1-3 of an fp32 or split-bf16 engine) against the two separate launches: same k order, same segment
    sums, same epilogue arithmetic -> same bits, on block-output taps and logits, including ragged last tiles
    (90x70: 414 / 108 pixels per frame) and the segmented-K layer2 convs.  bf16: the weight-stationary form
    (conv3x3_ws_kernel<true>, layer1.1-2 only; the mid tensor stays in registers); 256x256 with 3 clips = 384 tiles,
    more than one per persistent workgroup."""
    from workoutdetector_amd.engine import TsmEngine, launch_trace
    x = make_input(80 + h, b, 8, h, w)
    got = {}
    for flag in ('1', '0'):
        monkeypatch.setenv('TSM_FUSE_CONV23', flag)
        monkeypatch.setenv('TSM_FUSE_BLOCK', '0')    # (bf16: the whole-block kernel would otherwise take layer1 from both arms)
        eng = TsmEngine(height=h, width=w, max_clips=b, state_dict=sd0, dtype=dtype)
        with launch_trace() as tr:
            got[flag] = [eng.run(None, {'input': x})[0]] + [eng.forward_tap(x, s) for s in ('layer1.1', 'layer1.2', 'layer2.3')]
        eng.close()
        # what RAN: bf16 -> conv3x3_ws_kernel<true> (layer1.1-2); fp32 / split-bf16 -> conv23_fused_kernel<64 | 128, X3>
        fused_kernels = ['conv3x3_ws_kernel<true>'] if dtype == 'bf16' else \
            ['conv23_fused_kernel<%d, %s>' % (c, 'true' if dtype == 'bf16x3' else 'false') for c in (64, 128)]
        for kern in fused_kernels:
            assert tr.ran(kern) == (flag == '1'), (flag, kern, sorted(set(tr.kernels)))
    monkeypatch.delenv('TSM_FUSE_BLOCK')
    for a, c in zip(got['1'], got['0']):
        assert np.array_equal(a, c)
    monkeypatch.delenv('TSM_FUSE_CONV23')
    # left to the tuner, the choice is timing-based and bit-neutral
    eng = TsmEngine(height=h, width=w, max_clips=b, state_dict=sd0, dtype=dtype)
    assert np.array_equal(eng.run(None, {'input': x})[0], got['0'][0])
    picked = sorted(k for k, v in eng.conv_tiles(b).items() if v.endswith('+conv3'))
    print(f'\n[{dtype} {h}x{w} b{b}: fused conv2+conv3 picked by the tuner for] {picked}')
    assert set(picked) <= {'layer1.1.conv2', 'layer1.2.conv2', 'layer2.1.conv2', 'layer2.2.conv2', 'layer2.3.conv2'}
    eng.close()
