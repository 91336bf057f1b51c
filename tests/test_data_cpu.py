"""libdb1_data.so (include/db1_data.h): the memory-mapped token store and the index builders against golden vectors produced by
the reference itself (its MMapIndexedDatasetBuilder / MMapIndexedDataset and its native helpers.cpp: tests/golden/make_golden.py data),
and, when oracle/_ref is present (build container), directly against the reference's compiled helpers on random inputs."""
import ctypes
import os
import re
import sys

import numpy as np
import torch
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
G = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def D():
    from bdm_db1_amd.build import build_data_lib
    build_data_lib()
    from bdm_db1_amd.data import indexed
    return indexed


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(os.path.join(G, "data_ingest.npz")))


def test_library_exports_every_declared_symbol(D):
    hdr = open(os.path.join(ROOT, "include", "db1_data.h")).read()
    names = sorted(set(re.findall(r"\b(db1_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 15
    lib = ctypes.CDLL(os.path.join(ROOT, "bdm_db1_amd", "libdb1_data.so"))
    for n in names:
        assert hasattr(lib, n), n


def test_token_store_reads_what_the_reference_wrote(D, gold):
    ds = D.MMapIndexedDataset(os.path.join(G, "data_fixture"))
    assert len(ds) == len(gold["store_sizes"]) and ds.dtype == np.uint16
    assert np.array_equal(ds.sizes, gold["store_sizes"]) and np.array_equal(ds.doc_idx, gold["store_doc_idx"])
    assert np.array_equal(np.concatenate([ds[i] for i in range(len(ds))]), gold["store_flat"])
    assert np.array_equal(ds.get(3, 10, 20), gold["store_get_3_10_20"])
    assert np.array_equal(ds.get(8, 5), gold["store_get_8_5"])
    sl = ds[2:6]
    assert [len(x) for x in sl] == list(gold["store_slice_2_6_lens"])
    assert np.array_equal(np.concatenate(sl), gold["store_slice_2_6_flat"])
    assert ds[2:2] == [] and D.MMapIndexedDataset.exists(os.path.join(G, "data_fixture"))
    with pytest.raises(D.Db1DataError):
        ds.get(len(ds))
    with pytest.raises(D.Db1DataError):
        ds.get(0, 3, 100)
    with pytest.raises(ValueError):
        ds[0:4:2]
    with pytest.raises(D.Db1DataError):
        D.MMapIndexedDataset(os.path.join(G, "no_such_prefix"))


def test_rejects_a_corrupt_index(D, tmp_path):
    raw = open(os.path.join(G, "data_fixture.idx"), "rb").read()
    for name, blob in (("magic", b"XXIDIDX\x00\x00" + raw[9:]), ("short", raw[:40]), ("dtype", raw[:17] + b"\x63" + raw[18:])):
        p = tmp_path / name
        (tmp_path / (name + ".idx")).write_bytes(blob)
        (tmp_path / (name + ".bin")).write_bytes(open(os.path.join(G, "data_fixture.bin"), "rb").read())
        with pytest.raises(D.Db1DataError):
            D.MMapIndexedDataset(str(p))


def test_index_builders_match_reference_golden(D, gold):
    for k in range(4):
        seq, epochs, tpe = (int(v) for v in gold[f"sample_args{k}"])
        got = D.build_sample_idx(gold["sample_sizes"], gold[f"sample_doc_idx{k}"], seq, epochs, tpe)
        assert got.dtype == np.int32 and np.array_equal(got, gold[f"sample_idx{k}"]), k
    for tn in (1, 5, 47):
        assert np.array_equal(D.build_rl_sample_idx(gold["rl_path_lengths"], tn), gold[f"rl_idx_tn{tn}"])
    for size in (1, 10, 1000):
        di, dsi = np.zeros(size, np.uint8), np.zeros(size, np.int64)
        D.build_blending_indices(di, dsi, gold["blend_weights"], len(gold["blend_weights"]), size)
        assert np.array_equal(di, gold[f"blend_index_{size}"]) and np.array_equal(dsi, gold[f"blend_sample_{size}"])
    with pytest.raises(D.Db1DataError):
        D.build_sample_idx(gold["sample_sizes"], gold["sample_doc_idx0"], 1, 1, 100)  # seq_length must exceed 1


def test_index_builders_against_compiled_reference_helpers(D):
    """the reference's own helpers.cpp, compiled by oracle/Makefile (present in the build container only)"""
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    if not os.path.isdir(ref_dir) or not any(f.startswith("helpers") for f in os.listdir(ref_dir)):
        pytest.skip("oracle/_ref/helpers*.so not built (needs /root/reference)")
    sys.path.insert(0, ref_dir)
    import helpers as ref
    rng = np.random.default_rng(5)
    for trial in range(20):
        ndoc = int(rng.integers(1, 60))
        sizes = rng.integers(1, 300, ndoc).astype(np.int32)
        epochs = int(rng.integers(1, 4))
        doc_idx = np.concatenate([rng.permutation(ndoc) for _ in range(epochs)]).astype(np.int32)
        seq = int(rng.integers(2, 128))
        tpe = int(sizes.sum())
        if tpe <= 1:
            continue
        assert np.array_equal(D.build_sample_idx(sizes, doc_idx, seq, epochs, tpe), np.array(ref.build_sample_idx(sizes, doc_idx, seq, epochs, tpe)))
        pl = rng.integers(2, 100, int(rng.integers(1, 50))).astype(np.int32)
        tn = int(rng.integers(1, 60))
        assert np.array_equal(D.build_rl_sample_idx(pl, tn), np.array(ref.build_rl_sample_idx(pl, tn)))
        nd = int(rng.integers(1, 9))
        w = rng.random(nd)
        w /= w.sum()
        size = int(rng.integers(1, 5000))
        a, b = np.zeros(size, np.uint8), np.zeros(size, np.int64)
        c, e = np.zeros(size, np.uint8), np.zeros(size, np.int64)
        D.build_blending_indices(a, b, w, nd, size)
        ref.build_blending_indices(c, e, w, nd, size, False)
        assert np.array_equal(a, c) and np.array_equal(b, e)


def test_c_abi_under_address_and_undefined_behaviour_sanitizers(tmp_path, gold):
    """the data library's C ABI compiled with -fsanitize=address,undefined and driven from C over the reference-written
    store, including every error path and a truncated index: any out-of-bounds access of the mapping aborts the driver"""
    import shutil
    import subprocess
    src = os.path.join(ROOT, "bdm_db1_amd", "csrc_host", "db1_data.cpp")
    drv = os.path.join(ROOT, "tests", "csrc", "data_sanitize_main.c")
    exe = str(tmp_path / "data_sanitize")
    obj = str(tmp_path / "main.o")
    flags = ["-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"]
    r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror"] + flags + ["-c", drv, "-o", obj], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run(["g++", "-std=c++17", "-Wall"] + flags + [src, obj, "-o", exe], capture_output=True, text=True)
    if r.returncode != 0 and "asan" in r.stderr.lower():
        pytest.skip("libasan is not installed on this box")
    assert r.returncode == 0, r.stderr[-3000:]
    prefix = os.path.join(G, "data_fixture")
    trunc = str(tmp_path / "trunc")
    blob = open(prefix + ".idx", "rb").read()
    open(trunc + ".idx", "wb").write(blob[:len(blob) - 9])
    shutil.copy(prefix + ".bin", trunc + ".bin")
    r = subprocess.run([exe, prefix, trunc], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1"))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert r.stdout.startswith("ok items=%d tokens=%d " % (len(gold["store_sizes"]), int(gold["store_sizes"].sum()))), r.stdout


def test_gpt_dataset_and_blendable_dataset_match_reference_golden(D, gold):
    """bdm_db1_amd.data.gpt_dataset.GPTDataset / BlendableDataset (src/data/gpt_dataset.py:86-448, blendable_dataset.py:30-72) over the
    fixture store: the three index arrays (RandomState draw order + the C++ sample-index builder) and every sample the reference served"""
    from bdm_db1_amd.data.gpt_dataset import GPTDataset, get_ltor_masks_and_position_ids
    from bdm_db1_amd.data.blendable_dataset import BlendableDataset
    g = dict(np.load(os.path.join(G, "gpt_dataset.npz")))
    store = D.MMapIndexedDataset(os.path.join(G, "data_fixture"))
    k = 0
    while f"case{k}/args" in g:
        seq, seed, eod_mask, ne, sep = (int(x) for x in g[f"case{k}/args"])
        ds = GPTDataset("t", "unused_prefix", g[f"case{k}/docs"], store, None, seq, seed, eos_token_id=3, eod_mask_loss=bool(eod_mask))
        assert ds.doc_idx.dtype == np.int32 and np.array_equal(ds.doc_idx, g[f"case{k}/doc_idx"])
        assert np.array_equal(ds.sample_idx, g[f"case{k}/sample_idx"]) and np.array_equal(ds.shuffle_idx, g[f"case{k}/shuffle_idx"])
        assert ds.shuffle_idx.dtype == g[f"case{k}/shuffle_idx"].dtype and len(ds) == ds.sample_idx.shape[0] - 1
        items = [ds[i] for i in range(len(ds))]
        assert all(type(x).__name__ == "NLPTaskInput" and x.text_seq.shape == (1, seq) and x.text_seq.dtype == torch.int32 for x in items)
        assert np.array_equal(np.concatenate([x.text_seq.numpy() for x in items]), g[f"case{k}/text_seq"])
        assert np.array_equal(np.concatenate([x.label.numpy() for x in items]), g[f"case{k}/label"])
        assert np.array_equal(np.concatenate([x.loss_mask.numpy() for x in items]), g[f"case{k}/loss_mask"])
        assert np.array_equal(items[0].position_id.numpy(), g[f"case{k}/position_id"])
        k += 1
    assert k == 3
    am, lm, pid = get_ltor_masks_and_position_ids(np.array([5, 3, 7]), 3, False, False, True)
    assert am.dtype == bool and am[0, 1] and not am[1, 0] and lm.tolist() == [1.0, 0.0, 1.0] and pid.dtype == np.int64
    dsets = [list(range(100, 110)), list(range(200, 205)), list(range(300, 303))]
    b = BlendableDataset(dsets, [0.5, 0.3, 0.2], global_batch_size=8)
    np.random.seed(5)
    assert np.array_equal(b.offset_in_batch, g["blend/offsets"]) and len(b) == int(g["blend/len"])
    assert np.array_equal(np.array([b[i] for i in range(40)]), g["blend/items"])
    b2 = BlendableDataset(dsets, [1.0, 1.0, 2.0])
    np.random.seed(6)
    assert np.array_equal(b2.offset_in_batch, g["blend2/offsets"]) and np.array_equal(np.array([b2[i] for i in range(12)]), g["blend2/items"])


def test_gpt_dataset_index_cache_files_round_trip(D, tmp_path):
    """cache_prefix: the reference's `<prefix>_<name>_indexmap_<ns>ns_<sl>sl_<seed>s_{doc,sample,shuffle}_idx.npy` files are written once and
    then loaded instead of rebuilt"""
    from bdm_db1_amd.data.gpt_dataset import GPTDataset
    store = D.MMapIndexedDataset(os.path.join(G, "data_fixture"))
    a = GPTDataset("train", "p", np.arange(11), store, None, 16, 1234, cache_prefix=str(tmp_path / "corpus"))
    files = sorted(os.listdir(tmp_path))
    assert len(files) == 3 and all(f.startswith("corpus_train_indexmap_") and "_16sl_1234s_" in f for f in files)
    b = GPTDataset("train", "p", np.arange(11), store, None, 16, 1234, cache_prefix=str(tmp_path / "corpus"))
    assert np.array_equal(a.shuffle_idx, b.shuffle_idx) and np.array_equal(a[0].text_seq.numpy(), b[0].text_seq.numpy())
