"""serde-compatible JSON state files (gsearch_amd/state.py) against what the reference documents about them."""
import os

from gsearch_amd import state as S


def test_parameters_json_matches_readme_listing(tmp_path):
    # README.md:164-168 lists the sample database's parameters.json at 180 bytes; the recommended build of README.md:69
    # (-k 16 -s 18000 -n 128 --ef 1600 --scale_modify_f 0.25 --algo optdens) serialises to exactly that size
    p = S.ProcessingParams(S.HnswParams(1_500_000, 1600, 128, 0.25), 16, 18000, "optdens", "dna", False)
    assert p.to_json() == ('{"hnsw":{"capacity":1500000,"ef":1600,"max_nb_conn":128,"scale_modification":0.25},'
                           '"sketch":{"kmer_size":16,"sketch_size":18000,"algo":"OPTDENS","data_t":"DNA"},"block_flag":false}')
    assert len(p.to_json()) == 180
    p.dump_json(tmp_path)
    assert os.path.getsize(tmp_path / "parameters.json") == 180
    q = S.ProcessingParams.reload_json(tmp_path)
    assert q.to_json() == p.to_json() and q.get_hnsw_params().get_max_nb_connection() == 128 and q.get_kmer_size() == 16


def test_processing_state_and_seqdict_roundtrip(tmp_path):
    st = S.ProcessingState(318000, 318000, 7200.0)
    assert st.to_json() == '{"nb_seq":318000,"nb_file":318000,"elapsed_t":7200.0}'
    st.dump_json(tmp_path)
    assert S.ProcessingState.reload_json(tmp_path).to_json() == st.to_json()
    d = S.SeqDict()
    d.append("/db/GCF_000001.fna.gz", "NZ_CP0001.1", 4379993)
    d.append('/db/with "quote".fna', "id2", 5)
    fn = str(tmp_path / "seqdict.json")
    d.dump(fn)
    text = open(fn).read()
    assert text.startswith('{"id":{"path":"/db/GCF_000001.fna.gz","fasta_id":"NZ_CP0001.1"},"len":4379993}{"id":')   # back to back, no separator
    r = S.SeqDict.reload_json(fn)
    assert r.items == d.items and r.get_nb_entries() == 2 and r[0][2] == 4379993


def test_floats_are_written_like_ryu():
    """serde_json prints floats through ryu: no '+', no zero padding in exponents, decimal notation inside [1e-5, 1e16) for f64 and
    [1e-6, 1e13) for f32, shortest round-trip digits of the TYPE (elapsed_t is an f32 upstream, files.rs:29)"""
    R = S._ryu
    assert [R(x) for x in (0.25, 1.0, 7200.0, 1e-5, 1e-6, 1.5e-7, 1e16, 1e15, 123456789012345680.0, 0.1 + 0.2, -2.5, 0.0)] == \
        ["0.25", "1.0", "7200.0", "0.00001", "1e-6", "1.5e-7", "1e16", "1000000000000000.0", "1.2345678901234568e17", "0.30000000000000004", "-2.5", "0.0"]
    assert R(0.1, f32=True) == "0.1" and R(7200.5, f32=True) == "7200.5" and R(16777217.0, f32=True) == "16777216.0"
    assert R(1e13, f32=True) == "1e13" and R(1e-6, f32=True) == "0.000001" and R(1e-7, f32=True) == "1e-7"
    assert S.ProcessingState(3, 2, 0.1).to_json() == '{"nb_seq":3,"nb_file":2,"elapsed_t":0.1}'
    assert S.ProcessingState(3, 2, 1234.5678).to_json() == '{"nb_seq":3,"nb_file":2,"elapsed_t":1234.5677}'          # f32(1234.5678) = 1234.5677...
    p = S.ProcessingParams(S.HnswParams(1, 2, 3, 1e-5), 16, 100, "optdens")
    assert '"scale_modification":0.00001}' in p.to_json()


def test_non_ascii_paths_roundtrip_in_utf8(tmp_path):
    d = S.SeqDict()
    d.append("/données/génome_α.fna", "id β", 7)
    fn = str(tmp_path / "seqdict.json")
    d.dump(fn)
    assert "génome_α".encode("utf-8") in open(fn, "rb").read()
    assert S.SeqDict.reload_json(fn).items == d.items
