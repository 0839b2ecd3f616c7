"""Host logic of the video statistics against goldens produced by RUNNING the reference's own
functions (tools/make_golden.py golden_celeb_stat): the interval JSON files must match byte for byte."""
import json
import os

import pytest

from conftest import GOLDEN
from vn_celeb_face_recognition_amd import statistics as st

with open(os.path.join(GOLDEN, "celeb_stat_ref.json")) as _f:
    REF = json.load(_f)
CSV = os.path.join(GOLDEN, "celeb_stat_tracker.csv")


@pytest.mark.parametrize("tag", ["dynamic_5_4", "dynamic_7_2", "fixed_16_3", "fixed_83_1"])
@pytest.mark.parametrize("frame", ["dict", "pandas"])
def test_interval_json_matches_reference_bytes(tmp_path, tag, frame):
    if frame == "pandas":
        pd = pytest.importorskip("pandas")
        df = pd.read_csv(CSV)
    else:
        df = st.read_tracker_csv(CSV)
    mode, arg, nap = tag.split("_")
    out = str(tmp_path / "t.json")
    fn = st.export_json_stat_dynamic_itv if mode == "dynamic" else st.export_json_stat_fixed_itv
    fn(df, out, int(arg), int(nap), "Unknown", log=False)
    assert open(out).read() == REF[tag]["text"]


def test_time_format_and_sampling_and_thresholds(tmp_path):
    for k, want in REF["hms"].items():
        assert st.convert_sec_to_max_time_quantity(float(k)) == want
    # -fidx 1 6 11 16 at 25 fps keeps 4 frames of every 25 (celeb_statistic.py:180-187; count is 1-based)
    kept = [c for c in range(1, 101) if st.frame_is_sampled(c, 25.0, [1, 6, 11, 16])]
    assert kept == [1, 6, 11, 16, 26, 31, 36, 41, 51, 56, 61, 66, 76, 81, 86, 91]
    assert st.build_thresholds('', 3, 0.7) == {"0": 0.7, "1": 0.7, "2": 0.7}
    p = tmp_path / "thr.json"
    p.write_text(json.dumps({"0": 0.5, "1": 0.9}))
    assert st.build_thresholds(str(p), 3, 0.7) == {"0": 0.5, "1": 0.9}


def test_tracker_rows_round_trip_and_missing_emotion_column(tmp_path):
    """tracker.csv rows as celeb_statistic.py:253-276 writes them; without an Emotion column (no --recog_emotion: the
    reference raises KeyError there) the statistics still run, with empty emotion lists."""
    path = tmp_path / "tracker.csv"
    with open(path, "w") as f:
        f.write(",".join(st.tracker_header(True)) + "\n")
        for i in range(6):
            f.write(st.tracker_row(0.04 * (i + 1), ["a", "Unknown"], i + 1, [[10, 20, 110, 220], [0, 0, 50, 50]], (1000, 2000), True))
    df = st.read_tracker_csv(str(path))
    assert df["Names"][0] == "['a', 'Unknown']" and df["Frame_idx"] == [str(i + 1) for i in range(6)]
    out = st.export_json_stat_dynamic_itv(df, str(tmp_path / "o.json"), 2, 3, "Unknown", log=False)
    assert list(out) == ["1", "2"] and list(out["1"]["celebrities"]) == ["a"]
    item = out["1"]["celebrities"]["a"][0]
    assert item["bbox"] == [0.005, 0.02, 0.055, 0.22] and item["emotions"] == [] and item["time"] == "0.0h:0.0m:0.04s"
    assert json.load(open(tmp_path / "o.json"))["2"]["interval"] == ["0.0h:0.0m:0.16s", "0.0h:0.0m:0.24s"]
