"""Host side of the hot path on CPU with a stub model: clip windows, transform, JSON schema, evaluation."""
import json
import os
import sys

import numpy as np
import pandas as pd
import pytest
import torch

from oracle import counting_oracle, transform_oracle
from tests._stub import StubModel, synthetic_video
from workoutdetector_amd import eval as tsm_eval
from workoutdetector_amd import inference_count as ic
from workoutdetector_amd.repcount import RepcountHelper
from workoutdetector_amd.transform import TestTransform, build_test_transform, crop_offsets, resized_hw


@pytest.mark.parametrize('h,w', [(360, 206), (272, 480), (256, 256), (300, 224), (225, 640)])
def test_transform_matches_oracle(h, w):
    x = torch.rand(3, 3, h, w) * 255
    for scale in (False, True):
        got = TestTransform(scale_255=scale)(x)
        want = transform_oracle.test_transform(x, scale_255=scale)
        assert tuple(got.shape) == (3, 3, 224, 224)
        assert torch.equal(got, want)
    assert resized_hw(h, w) == transform_oracle.resized_hw(h, w)
    assert crop_offsets(*resized_hw(h, w)) == transform_oracle.crop_offsets(*transform_oracle.resized_hw(h, w))


def test_person_crop_is_out_of_scope():
    with pytest.raises(NotImplementedError):
        build_test_transform(person_crop=True)


@pytest.mark.parametrize('frames', [1, 7, 8, 9, 16, 17, 100, 336])
def test_clip_windows_match_reference_loop(frames):
    """range(0, F, 8) x vid[i:i+16:2], tail zero-padded, float32 0..255 (utils/inference_count.py:411-414)."""
    vid = torch.from_numpy(synthetic_video(frames, frames, 12, 10))
    starts = ic.clip_starts(frames)
    assert starts == counting_oracle.clip_starts(frames) and len(starts) == -(-frames // 8)
    for s in starts:
        clip = ic.make_clip(vid, s)
        assert torch.equal(clip, transform_oracle.make_clip(vid, s))
        idx = counting_oracle.clip_frame_indices(frames, s)
        assert torch.equal(clip[:len(idx)], vid[idx].float())
        assert float(clip[len(idx):].abs().sum()) == 0.0


@pytest.mark.parametrize('frames,h,w', [(50, 60, 40), (8, 40, 64), (131, 48, 48)])
def test_batched_video_path_equals_per_clip_reference_path(frames, h, w):
    """Transforming each even frame once and gathering windows == the reference's clip-by-clip loop
    (make_clip -> inference_video) on the same model, bit for bit."""
    vid = torch.from_numpy(synthetic_video(3, frames, h, w))
    model = StubModel()
    tf = build_test_transform(False)
    batched = ic.video_clip_logits(model, vid, tf, batch_clips=5)
    per_clip = []
    for s in ic.clip_starts(frames):
        pred = ic.inference_video(model, ic.make_clip(vid, s), transform=tf)
        assert [c for c, _ in pred] == list(range(12))              # unsorted enumerate, like the reference
        per_clip.append([v for _, v in pred])
    assert np.array_equal(batched.numpy(), np.array(per_clip, dtype=np.float32))
    # sub-ranges (what a rank computes under sharding) are slices of the full result
    n = len(per_clip)
    for lo, hi in [(0, 1), (1, n), (n // 2, n), (n - 1, n)]:
        if lo < hi:
            assert np.array_equal(ic.video_clip_logits(model, vid, tf, (lo, hi)).numpy(), batched[lo:hi].numpy())


@pytest.fixture
def tiny_dataset(tmp_path, golden_dir):
    """Three RepCount test videos (names and ground truth from annotation.csv) as synthetic .npy frames."""
    anno = pd.read_csv(f'{golden_dir}/repcount_annotation.csv', index_col=0)
    names = ['stu1_40.mp4', 'stu5_32.mp4', 'stu3_53.mp4']
    rows = anno[anno['name'].isin(names)].copy()
    assert len(rows) == 3
    rows['name'] = [n.replace('.mp4', '.npy') for n in rows['name']]
    root = tmp_path / 'RepCount'
    (root / 'videos' / 'test').mkdir(parents=True)
    rows.to_csv(root / 'annotation.csv')
    frames = {'stu1_40.npy': 84, 'stu5_32.npy': 131, 'stu3_53.npy': 40}
    for i, name in enumerate(rows['name']):
        np.save(root / 'videos' / 'test' / name, synthetic_video(i, frames[name], 45, 26, period=20 + 4 * i))
    return str(root)


def test_inference_dataset_schema_and_eval(tiny_dataset, tmp_path):
    out_dir = str(tmp_path / 'out')
    model = StubModel()
    ic.inference_dataset(model, ['test'], out_dir, checkpoint='stub', data_root=tiny_dataset)
    files = sorted(os.listdir(out_dir))
    assert files == ['stu1_40.npy.score.json', 'stu3_53.npy.score.json', 'stu5_32.npy.score.json']
    d = json.load(open(os.path.join(out_dir, 'stu1_40.npy.score.json')))
    assert list(d) == ['video_name', 'model', 'input_shape', 'checkpoint', 'total_frames', 'ground_truth', 'action',
                       'scores']
    assert d['model'] == 'video_model' and d['input_shape'] == [1, 8, 3, 224, 224] and d['checkpoint'] == 'stub'
    assert d['total_frames'] == 84 and d['action'] == 'pull_up' and len(d['ground_truth']) == 16
    assert list(d['scores']) == [str(i) for i in range(0, 84, 8)]       # int keys -> str through JSON
    assert list(d['scores']['0']) == [str(c) for c in range(12)]
    # scores -> states -> counts, product vs oracle, with and without softmax
    for softmax in (False, True):
        preds = tsm_eval.preds_from_scores(d['scores'], softmax=softmax)
        rows = [[d['scores'][k][str(c)] for c in range(12)] for k in d['scores']]
        assert preds == counting_oracle.scores_to_preds(rows, use_softmax=softmax)
    # eval main over the directory: rename so that names map back to annotation rows ('x.npy.score.json' -> 'x.mp4')
    anno = os.path.join(tiny_dataset, 'annotation_mp4.csv')
    a = pd.read_csv(os.path.join(tiny_dataset, 'annotation.csv'), index_col=0)
    a['name'] = [n.replace('.npy', '.mp4') for n in a['name']]
    a.to_csv(anno)
    mae, obo = tsm_eval.main(out_dir, anno, str(tmp_path / 'eval.csv'), softmax=True)
    df = pd.read_csv(tmp_path / 'eval.csv')
    assert len(df) == 3 and set(df['action']) <= {'pull_up', 'squat', 'situp', 'push_up', 'jump_jack', 'front_raise'}
    want = counting_oracle.obo_mae(list(df['pred_count']), list(df['gt_count']))
    assert (mae, obo) == want
    assert len(tsm_eval.summarize_counts(df)) >= 1
    # analyze_count(csv, out_csv): the reference's signature; action 'all' rows close every split
    tsm_eval.analyze_count(str(tmp_path / 'eval.csv'), str(tmp_path / 'eval_meta.csv'))
    meta = pd.read_csv(tmp_path / 'eval_meta.csv', index_col=0)
    assert list(meta.columns) == ['action', 'split', 'mae', 'obo_acc', 'total', 'avg_count']
    assert int(meta[meta.action == 'all'].total.sum()) == 3


def test_streaming_counter_matches_offline(tmp_path):
    """count_by_video_model: non-overlapping 8-frame windows, identical to thresholding + pred_to_count."""
    vid = synthetic_video(5, 90, 30, 40, period=16)
    model = StubModel(gain=8.0)
    seen = []
    count, reps = ic.count_by_video_model(model, iter(vid), on_window=lambda i, s, c: seen.append((i, s, c)))
    tf = build_test_transform(False)
    states = []
    for i in range(0, 88, 8):
        clip = torch.from_numpy(vid[i:i + 8]).float()
        scores = [v for _, v in ic.inference_video(model, clip, transform=tf)]
        states.append(counting_oracle.scores_to_preds([scores])[0])
    assert [s for _, s, _ in seen] == states and len(states) == 11      # last 2 frames never fill a window
    assert (count, reps) == counting_oracle.pred_to_count(states, 8)


def test_repcount_helper_against_annotation(golden_dir):
    h = RepcountHelper('/nonexistent', f'{golden_dir}/repcount_annotation.csv')
    val, test = h.get_rep_data(['val'], ['all']), h.get_rep_data(['test'], ['all'])
    assert len(val) == 100 and len(test) == 117
    it = test['stu1_40.mp4']
    assert (it.count, it.class_, it.split, it.reps[:2], len(it.reps)) == (8, 'pull_up', 'test', [19, 54], 16)
    assert it.video_path == '/nonexistent/videos/test/stu1_40.mp4' and it.total_frames == -1
    assert all(i.class_ != 'bench_pressing' for i in test.values())
    pull = h.get_rep_data(['test'], ['pull_up'])
    assert set(pull) <= set(test) and all(i.class_ == 'pull_up' for i in pull.values())
    # eval_count property from the reference's tests/test_repcount_dataset.py:66-85
    gt = {k: v.count for k, v in pull.items()}
    mae, obo, per = h.eval_count({k: c + 1 for k, c in gt.items()}, ['test'], ['pull_up'])
    assert obo == 1.0 and mae == pytest.approx(np.mean([1 / c if c else 0 for c in gt.values()]))


def test_save_scores_to_json_refuses_overwrite(tmp_path):
    p = str(tmp_path / 'a')
    ic.save_scores_to_json([[0.1, 0.9], [0.8, 0.2]], p, 'v.mp4', step=8)
    d = json.load(open(p + '.json'))
    assert d['scores'] == {'0': {'0': 0.1, '1': 0.9}, '8': {'0': 0.8, '1': 0.2}}
    with pytest.raises(AssertionError):
        ic.save_scores_to_json([[0.0, 1.0]], p, 'v.mp4', step=8)


def test_read_video_npy_and_missing_decoder(tmp_path):
    v = synthetic_video(0, 5, 8, 8)
    np.save(tmp_path / 'v.npy', v)
    assert torch.equal(ic.read_video(str(tmp_path / 'v.npy')), torch.from_numpy(v))
    np.save(tmp_path / 'bad.npy', v.astype(np.float32))
    with pytest.raises(ValueError):
        ic.read_video(str(tmp_path / 'bad.npy'))
    try:
        import torchvision  # noqa: F401
    except ImportError:
        with pytest.raises(RuntimeError, match='no video decoder'):
            ic.read_video('/nonexistent/video.mp4')


def test_stream_batcher_matches_per_stream_counting():
    """Many interleaved streams through one batched model call == each stream counted on its own
    (count_by_video_model); frame sizes differ between streams; max_batch splits a step."""
    from workoutdetector_amd.streaming import StreamBatcher
    vids = {'a': synthetic_video(1, 90, 30, 40, period=16), 'b': synthetic_video(2, 61, 36, 36, period=20),
            'c': synthetic_video(3, 40, 30, 40, period=12)}
    model = StubModel(gain=8.0)
    sb = StreamBatcher(model, max_batch=3)
    events = {k: [] for k in vids}
    for t in range(90):                                  # frames arrive interleaved; step every 10 ticks
        for k, v in vids.items():
            if t < len(v):
                sb.push(k, v[t])
        if t % 10 == 9:
            for k, ev in sb.step().items():
                events[k] += ev
    for k, ev in sb.step().items():
        events[k] += ev
    assert sb.ready() == 0 and model.calls < sum(len(v) // 8 for v in vids.values())    # windows were batched
    for k, v in vids.items():
        seen = []
        want = ic.count_by_video_model(StubModel(gain=8.0), iter(v), on_window=lambda i, s, c: seen.append((i, s, c)))
        assert events[k] == seen
        assert sb.result(k) == want and sb.close(k) == want
    assert not sb.streams
    with pytest.raises(ValueError):
        sb.push('d', np.zeros((4, 4, 3), np.float32))


def test_clip_batcher_hands_every_row_back_to_its_video():
    """_ClipBatcher (full batches across video boundaries, the dataset loop of shard='global'): random video lengths --
    empty ones, single clips, lengths below / equal to / far above the batch -- and random batch sizes; every video must
    get exactly its own clips' rows, in order, and the model must see ceil(total / batch) calls (full batches)."""
    rng = np.random.default_rng(7)

    class Echo:                                # logits row = [video key, clip index, first pixel of the clip]
        num_class = 3

        def __init__(self):
            self.calls = 0

        def get_inputs(self):
            from tests._stub import _Arg
            return [_Arg('input')]

        def run(self, _names, feed):
            (x,) = feed.values()
            self.calls += 1
            return [np.stack([x[:, 0, 0, 0, 0], x[:, 0, 0, 0, 1], x[:, 0, 0, 0, 2]], axis=1).astype(np.float32)]

    for _ in range(25):
        batch = int(rng.integers(1, 9))
        counts = [int(c) for c in rng.choice([0, 1, 2, 3, batch - 1, batch, batch + 1, 3 * batch + 2, 17], size=int(rng.integers(1, 7)))]
        model = Echo()
        batcher = ic._ClipBatcher(model, batch)
        for key, n in enumerate(counts):
            if n == 0:
                batcher.rows.setdefault(key, [])
                continue
            # "frames": one distinguishable frame per clip; clip i gathers frame i eight times
            frames = torch.zeros(n, 3, 2, 4)
            frames[:, 0, 0, 0] = key
            frames[:, 0, 0, 1] = torch.arange(n, dtype=torch.float32)
            frames[:, 0, 0, 2] = torch.arange(n, dtype=torch.float32) * 0.5 + key
            idx = torch.arange(n)[:, None].repeat(1, 8)
            batcher.add(key, frames, idx, False)
        batcher.flush()
        for key, n in enumerate(counts):
            rows = batcher.logits(key)
            assert tuple(rows.shape)[0] == n
            if n:
                want = torch.stack([torch.full((n,), float(key)), torch.arange(n, dtype=torch.float32),
                                    torch.arange(n, dtype=torch.float32) * 0.5 + key], dim=1)
                assert torch.equal(rows.to(torch.float32), want), (counts, batch, key)
        assert model.calls == -(-sum(counts) // batch), (counts, batch, model.calls)


def test_estimated_clips_sources(tmp_path):
    """The shard plan's clip estimate: a frame_counter callback, else the rawframes count, else a .npy header, else the
    last annotated repetition frame -- never a decode."""
    from workoutdetector_amd.repcount import RepcountItem
    item = RepcountItem(str(tmp_path / 'v.npy'), str(tmp_path / 'raw'), -1, 'squat', 2, [3, 40, 41, 97], 'test', 'v.npy')
    assert ic.estimated_clips(item) == len(range(0, 97, 8))                       # annotation only: 13 clips
    np.save(tmp_path / 'v.npy', np.zeros((150, 4, 4, 3), dtype=np.uint8))
    assert ic.estimated_clips(item) == len(range(0, 150, 8))                      # the .npy header: 19
    item.total_frames = 64
    assert ic.estimated_clips(item) == 8                                          # a rawframes directory wins
    assert ic.estimated_clips(item, frame_counter=lambda p: 9) == 2               # a callback wins over everything
    empty = RepcountItem('nowhere.mp4', '', -1, 'squat', 0, [], 'test', 'nowhere.mp4')
    assert ic.estimated_clips(empty) == 1


def test_bench_names_the_kernels_rocprof_prints():
    """bench.kernel_of maps the tuner's tile names to kernel names; roofline.kernel / roofline.traffic are only right if
    those are the names rocprofv3 prints.  Checked against the committed kernel-stats summaries of this round."""
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seen = {}
    for mode in ('f32', 'bf16c5', 'bf16x3'):
        text = open(os.path.join(root, 'profiles', f'r04_{mode}_kernel_stats.csv')).read()
        seen[mode] = text
    cases = [('f32', '64x64', 'f32', 256), ('f32', '64x64+conv3', 'f32', 128), ('f32', '64x64+conv3', 'f32', 64),
             ('bf16c5', '256x256', 'bf16', 256), ('bf16c5', '256x256p', 'bf16', 256), ('bf16c5', 'ws', 'bf16', 128), ('bf16x3', '128x128w8', 'bf16x3', 256),
             ('bf16x3', '128x128+conv3', 'bf16x3', 128)]
    for mode, tile, dtype, cmid in cases:
        name, with_conv3 = bench.kernel_of(tile, dtype, cmid)
        assert with_conv3 == tile.endswith('+conv3')
        assert ('tsm::' + name) in seen[mode], (tile, dtype, cmid, name)
    name, _ = bench.kernel_of('ws', 'bf16', 128, 2)           # layer2.0's conv2: the stride-2 arm
    assert name == 'conv3x3_ws128_kernel<true>' and ('tsm::' + name) in seen['bf16c5']
    assert bench.kernel_of('ws+conv3', 'bf16', 64) == ('conv3x3_ws_kernel<true>', True)
    assert bench.kernel_of('64x64/splitK', 'f32', 512)[0] == 'conv_igemm<64, 64, 2, 2, 3, false, false, 0, false, true>'


def test_profiling_tools_find_the_forwards_with_and_without_a_pack_launch():
    """tools/hbm_traffic.forward_starts delimits the forwards of a kernel trace: by the pack launch of an NTCHW input, or --
    since the pool-fused stem reads that layout itself -- by a stem launch that no pack precedes; tools/layer_times matches
    a forward's launches to layers whether or not conv3 of a block also ran the next block's conv1."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'tools'))
    try:
        from hbm_traffic import forward_starts
        from layer_times import match_schedule
    finally:
        sys.path.pop(0)
    fwd_pack = ['tsm::pack_input_kernel<2>', 'tsm::stem_pool_kernel<false, false>', 'tsm::bneck_ws_kernel<64, true, false>', 'tsm::head_fc_kernel']
    fwd_planar = ['tsm::stem_pool_kernel<false, true>', 'tsm::bneck_ws_kernel<64, true, false>', 'tsm::head_fc_kernel']
    assert forward_starts(fwd_pack * 3) == [0, 4, 8]
    assert forward_starts(fwd_planar * 3) == [0, 3, 6]
    assert forward_starts(fwd_pack + fwd_planar + fwd_pack) == [0, 4, 7]
    # one bf16 forward as the engine launches it at config 5: whole-block layer1, cross-block launches in layer2-3
    names = ['stem_pool_kernel<false, true>'] + ['bneck_ws_kernel<64, true, false>'] + ['bneck_ws_kernel<256, true, true>'] * 2
    names += ['conv1x1_wsn_kernel<256, 128, false>', 'conv3x3_ws128_kernel<true>', 'conv_bf16_256p_kernel<1, false, false, true>']      # layer2.0
    names += ['conv1x1_wsn_kernel<512, 128, false>', 'conv3x3_ws128_kernel<false>', 'conv31_fused_kernel<128, 512, 128, 1>']           # layer2.1 (+ 2.2.conv1)
    names += ['conv3x3_ws128_kernel<false>', 'conv31_fused_kernel<128, 512, 128, 1>', 'conv3x3_ws128_kernel<false>', 'conv31_fused_kernel<128, 512, 256, 2>']
    names += ['conv_bf16_256p_kernel<3, false, false, false>', 'conv_bf16_256p_kernel<1, false, false, true>']                          # layer3.0 (conv1 came fused)
    names += ['conv_bf16_256p_kernel<1, true, false, false>'] + ['conv_bf16_256p_kernel<3, false, false, false>', 'conv31_fused_kernel<256, 1024, 256, 2>'] * 4
    names += ['conv_bf16_256p_kernel<3, false, false, false>', 'conv_bf16_256p_kernel<1, false, true, false>']                           # layer3.5
    names += ['conv_bf16_256p_kernel<1, true, false, false>', 'conv_bf16_256p_kernel<3, false, false, false>', 'conv_bf16_256p_kernel<1, false, false, true>']
    names += ['conv_bf16_256p_kernel<1, true, false, false>', 'conv_bf16_256p_kernel<3, false, false, false>', 'conv_bf16_256p_kernel<1, false, true, false>'] * 2
    rows, ok = match_schedule([dict(Kernel_Name='void tsm::' + n + '(tsm::ConvParams)') for n in names])
    assert ok and len(rows) == len(names)
    labels = [r[0] for r in rows]
    assert labels[0] == 'conv1' and labels[1] == 'layer1.0 (block)' and labels[4] == 'layer2.0.conv1'
    assert 'layer2.1.conv3+layer2.2.conv1' in labels and 'layer2.3.conv3+layer3.0.conv1' in labels and 'layer3.4.conv3+layer3.5.conv1' in labels
    assert labels[-1] == 'layer4.2.conv3' and 'layer3.0.conv1' not in labels and 'layer3.1.conv1' in labels
    # round 5's schedule: layer2.0's conv1 + stride-2 conv2 as one launch, its conv3 + downsample on workgroup pairs, the
    # producer / consumer form of conv3 + next conv1
    names5 = list(names)
    names5[4:7] = ['front_s2_kernel<true>', 'conv1x1_wsn_kernel<384, 256, true, 2>']
    names5 = [n.replace('conv31_fused_kernel<128, 512, 256, 2>', 'conv31_pc_kernel<128, 512, 256>')
               .replace('conv31_fused_kernel<256, 1024, 256, 2>', 'conv31_pc_kernel<256, 1024, 256>') for n in names5]
    rows5, ok5 = match_schedule([dict(Kernel_Name='void tsm::' + n + '(tsm::ConvParams)') for n in names5])
    assert ok5 and len(rows5) == len(names5) == len(names) - 1
    labels5 = [r[0] for r in rows5]
    assert labels5[4] == 'layer2.0.conv1+conv2' and rows5[4][1] == ['layer2.0.conv1', 'layer2.0.conv2']
    assert labels5[5] == 'layer2.0.conv3+downsample' and 'layer3.4.conv3+layer3.5.conv1' in labels5 and labels5[-1] == 'layer4.2.conv3'


def test_per_launch_roofline_columns_of_the_traffic_table(tmp_path, capsys):
    """tools/traffic_per_launch.py: with a kernel trace of the same schedule every launch gets us / bound_us / x_bound / roof --
    bound = max(algorithmic bytes / 8 TB/s, algorithmic flops / the mode's dense MFMA peak).  A two-launch synthetic forward
    (the stem and a whole layer1.0 block) is priced by hand here."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'tools'))
    try:
        import traffic_per_launch as tpl
    finally:
        sys.path.pop(0)
    kernels = ['void tsm::stem_pool_kernel<false, true>(float const*)', 'void tsm::bneck_ws_kernel<64, true, false>(tsm::BneckParams)',
               'void tsm::head_pool_kernel<2>(float const*)']

    def counter_csv(d, counter, values):
        os.makedirs(d)
        with open(os.path.join(d, 'run_counter_collection.csv'), 'w') as f:
            f.write('Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value\n')
            disp = 1
            for rep in range(3):           # three forwards: the tools take the last whole one
                for k, v in zip(kernels, values):
                    f.write(f'{disp},"{k}",{counter},{v}\n')
                    disp += 1
    frames, size = 256, 224
    # KiB units; the read counter is doubled by the tool (gfx950 note)
    counter_csv(str(tmp_path / 'f'), 'FETCH_SIZE', [400, 900, 1])
    counter_csv(str(tmp_path / 'w'), 'WRITE_SIZE', [256, 4096, 1])
    trace = tmp_path / 'trace.csv'
    with open(trace, 'w') as f:
        f.write('Kernel_Name,Start_Timestamp,End_Timestamp,Grid_Size_X,Workgroup_Size_X\n')
        t = 0
        for rep in range(3):
            for k, us in zip(kernels, (500.0, 800.0, 1.0)):
                f.write(f'"{k}",{t},{t + int(us * 1000)},256,256\n')
                t += int(us * 1000) + 100
    tpl.main(str(tmp_path / 'f'), str(tmp_path / 'w'), frames, size, 2, str(trace))
    out = capsys.readouterr().out.splitlines()
    head = out[0].split()
    assert head[-4:] == ['us', 'bound_us', 'x_bound', 'roof']
    from workoutdetector_amd.flops import layer_table
    by = {r['name']: r for r in layer_table(size, size)}
    row = next(l for l in out if l.startswith('layer1.0 (block)')).split()
    us, bound, xb, roof = float(row[-4]), float(row[-3]), float(row[-2]), row[-1]
    flops = 2.0 * frames * sum(by[q]['macs'] for q in ('layer1.0.conv1', 'layer1.0.conv2', 'layer1.0.conv3', 'layer1.0.downsample'))
    hw = (size // 4) ** 2
    # the block input for conv1 and once more for the downsample branch (the kernel does read it twice), the output, the four packed weight matrices
    alg = frames * hw * (64 + 64 + 256) * 2 + (64 * 64 + 64 * 64 * 9 + 256 * 64 + 256 * 64) * 2
    want = max(alg / 8.0e12, flops / 2.5e15) * 1e6
    assert us == 800.0 and abs(bound - want) < 0.06 and abs(xb - 800.0 / want) < 0.01 and roof == ('hbm' if alg / 8.0e12 >= flops / 2.5e15 else 'mfma')
    assert any(l.startswith('conv-like launches: 1300 us against') for l in out)


def test_prefetch_pieces_covers_every_clip_once_and_survives_failures():
    """prefetch_pieces (the dataset loop's stager): every video comes as consecutive clip ranges of at most piece_clips
    that cover its clips exactly once, the last one flagged; a reader that raises surfaces in the consumer; a consumer
    that walks away does not leave the worker blocked on a full queue."""
    import threading
    import time
    from tests._stub import StubModel, synthetic_video
    from workoutdetector_amd import inference_count as ic
    lens = [77, 8, 1, 130, 16, 0]
    vids = [(k, (lambda i=i, n=n: torch.from_numpy(synthetic_video(i, n, 20, 18)) if n else torch.zeros((0, 20, 18, 3), dtype=torch.uint8)))
            for k, (i, n) in enumerate(enumerate(lens))]
    seen = {}
    for key, st, last in ic.prefetch_pieces(StubModel(), vids, piece_clips=4, depth=2):
        a = seen.setdefault(key, dict(next=0, last=False, total=st.total))
        assert not a['last'] and st.lo == a['next'] and st.hi - st.lo <= 4 and st.total == lens[key]
        a['next'], a['last'] = st.hi, last
    assert sorted(seen) == list(range(len(lens)))
    for k, n in enumerate(lens):
        assert seen[k]['last'] and seen[k]['next'] == len(ic.clip_starts(n)), (k, seen[k])

    def bad():
        raise OSError('no such video')
    with pytest.raises(OSError, match='no such video'):
        for _ in ic.prefetch_pieces(StubModel(), [(0, vids[0][1]), (1, bad)], piece_clips=4):
            pass
    # early exit: the generator is closed with the queue full; the worker must come home
    before = {t.name for t in threading.enumerate()}
    gen = ic.prefetch_pieces(StubModel(), vids, piece_clips=1, depth=1)
    next(gen)
    gen.close()
    deadline = time.time() + 5
    while any(t.name == 'tsm-stage' and t.is_alive() for t in threading.enumerate()) and time.time() < deadline:
        time.sleep(0.05)
    assert not any(t.name == 'tsm-stage' and t.is_alive() for t in threading.enumerate()), before


def test_gpu_gaps_tool_accounts_for_overlap_and_idle_time(tmp_path, capsys):
    """tools/gpu_gaps.py (the config-4 idle-time analysis behind profiles/r03_config4_gpu_gaps*.txt): busy time is the
    UNION of the kernels' intervals (concurrent kernels on two streams are not counted twice) and every idle interval
    is attributed to the kernel that ended it."""
    import csv
    import runpy
    trace = tmp_path / 'trace.csv'
    rows = [(0, 1_000_000, 'void tsm::preprocess_kernel<unsigned char>(tsm::PreprocParams)'),
            (500_000, 1_500_000, 'void tsm::gather_clips_kernel(tsm::GatherParams)'),                 # overlaps the first
            (1_500_010, 2_500_000, 'void tsm::conv_igemm<64, 64, 2, 2, 3, false, false, 0, false, true>(tsm::ConvParams)'),
            (4_500_000, 5_000_000, 'void at::native::(anonymous namespace)::indexSelectSmallIndex<float, long>(x)')]  # 2 ms idle before it
    with open(trace, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['Kernel_Name', 'Start_Timestamp', 'End_Timestamp'])
        for s, e, n in rows:
            w.writerow([n, s, e])
    mod = runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'gpu_gaps.py'))
    mod['main'](str(trace), 1000.0)
    out = capsys.readouterr().out
    assert 'span 5.0 ms, busy 3.0 ms (0.600), idle 2.0 ms in 2 gaps' in out, out
    assert 'gaps >= 1000 us: 1, 2.0 ms' in out and 'ended by at::native::indexSelectSmallIndex<float, long>' in out, out


def test_a_failing_forward_ends_the_dataset_job_without_a_stuck_stager(tiny_dataset, tmp_path):
    """An exception inside the dataset loop (here: the session's second forward) surfaces to the caller and the stager
    thread, which may be blocked on its full queue at that moment, goes home."""
    import threading
    import time

    class Flaky(StubModel):
        def run(self, output_names, feed):
            if self.calls >= 1:
                raise RuntimeError('device lost')
            return super().run(output_names, feed)

    with pytest.raises(RuntimeError, match='device lost'):
        ic.inference_dataset(Flaky(), ['test'], str(tmp_path / 'out'), checkpoint='stub', data_root=tiny_dataset, batch_clips=4)
    deadline = time.time() + 5
    while any(t.name == 'tsm-stage' and t.is_alive() for t in threading.enumerate()) and time.time() < deadline:
        time.sleep(0.05)
    assert not any(t.name == 'tsm-stage' and t.is_alive() for t in threading.enumerate())
