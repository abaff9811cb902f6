"""Loading granne's files on the GPU: Granne::from_file / from_bytes (src/index/mod.rs:106-135)
over Vectors::from_file, and save_index / save_elements (py/src/lib.rs:318-343, 509-535).
cf. the reference's write_and_load / write_and_load_compressed (src/index/tests.rs:337-451)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import fileformat as off  # noqa: E402
from tests.conftest import random_floats  # noqa: E402


@pytest.mark.parametrize("int8", [False, True])
def test_load_files_and_search(oracle, tmp_path, int8):
    import granne_amd
    rng = np.random.default_rng(21 + int8)
    raw = random_floats(rng, 1500, 100)
    el = oracle.quantize(raw) if int8 else oracle.normalize_f32(raw)
    oix = oracle.build_index(el, num_neighbors=20, max_search=20, n_threads=0)
    ip, ep = tmp_path / "index.granne", tmp_path / "elements.bin"
    ip.write_bytes(off.write_index(oix.layers))       # files as the reference would write them
    ep.write_bytes(off.write_elements(el))
    et = "angular_int" if int8 else "angular"
    gix = granne_amd.Granne.from_files(str(ip), et, str(ep))
    assert len(gix) == 1500 and gix.num_layers() == len(oix.layers) and gix.dim == 100
    for l, layer in enumerate(oix.layers):
        assert gix.layer_len(l) == layer.shape[0]
        want = sorted(int(x) for x in layer[3] if x != oracle.UNUSED)
        assert gix.get_neighbors(3, l) == want      # loaded layers hold sorted ids (Layers::Compressed)
    q = oracle.quantize(random_floats(rng, 40, 100)) if int8 else oracle.normalize_f32(random_floats(rng, 40, 100))
    ids, ds, cnt = gix.search_batch(q, 50, 10)
    oi, od, oc, _ = oix.search_batch(q, 50, 10)
    assert (ids == oi).all() and ds.tobytes() == od.tobytes() and (cnt == oc).all()
    # from_bytes
    g2 = granne_amd.Granne.from_bytes(ip.read_bytes(), et, ep.read_bytes())
    assert (g2.search_batch(q, 50, 10)[0] == oi).all()
    # save and reload: the written files are byte-identical to the reference-format files above
    gix.save_index(str(tmp_path / "i2.granne"))
    gix.save_elements(str(tmp_path / "e2.bin"))
    assert (tmp_path / "i2.granne").read_bytes() == ip.read_bytes()
    assert (tmp_path / "e2.bin").read_bytes() == ep.read_bytes()


def test_builder_save_and_load(oracle, tmp_path):
    import granne_amd
    rng = np.random.default_rng(23)
    el = oracle.normalize_f32(random_floats(rng, 1200, 28))
    b = granne_amd.GranneBuilder("angular", el, num_neighbors=20, max_search=20, batch_max=64)
    b.build()
    b.save_index(str(tmp_path / "b.granne"))
    b.save_elements(str(tmp_path / "b.bin"))
    meta, layers = off.read_index((tmp_path / "b.granne").read_bytes())
    assert meta["num_elements"] == 1200 and meta["layer_counts"] == [b.layer_len(l) for l in range(b.num_layers())]
    for l, nodes in enumerate(layers):
        got = b.get_layer(l)
        for i in (0, len(nodes) // 2, len(nodes) - 1):
            assert sorted(int(x) for x in got[i] if x != oracle.UNUSED) == nodes[i]
    assert (off.read_elements((tmp_path / "b.bin").read_bytes(), np.float32) == el).all()
    gix = granne_amd.Granne.from_files(str(tmp_path / "b.granne"), "angular", str(tmp_path / "b.bin"))
    direct = b.get_index()
    q = oracle.normalize_f32(random_floats(rng, 16, 28))
    a, c = gix.search_batch(q, 30, 5), direct.search_batch(q, 30, 5)
    assert (a[0] == c[0]).all() and a[1].tobytes() == c[1].tobytes()


def test_bad_files(tmp_path):
    import granne_amd
    (tmp_path / "x").write_bytes(b"not an index" + b" " * 1100)
    (tmp_path / "e").write_bytes(off.write_elements(np.zeros((4, 8), np.float32)))
    with pytest.raises(granne_amd.GranneHipError) as e:
        granne_amd.Granne.from_files(str(tmp_path / "x"), "angular", str(tmp_path / "e"))
    assert e.value.code == -5
    with pytest.raises(granne_amd.GranneHipError):
        granne_amd.Granne.from_files(str(tmp_path / "missing"), "angular", str(tmp_path / "e"))
