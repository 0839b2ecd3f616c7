"""Host-side mirror of the reference's video statistics: same names, argument meaning and
on-disk formats as /root/reference/celeb_statistic.py and utils/utils.py.

  convert_sec_to_max_time_quantity   <- utils/utils.py:77-82
  find_celeb_infor_in_interval       <- celeb_statistic.py:79-107
  export_json_stat_dynamic_itv       <- celeb_statistic.py:32-53
  export_json_stat_fixed_itv         <- celeb_statistic.py:56-76
  frame_is_sampled                   <- celeb_statistic.py:180-187   (-fidx: frames kept per second of video)
  build_thresholds                   <- celeb_statistic.py:127-136   (local_thresholds.json or one global value)
  tracker_row / tracker_header       <- celeb_statistic.py:137-147, 253-276 (tracker.csv)

`tracker_df` is anything with column access by name returning sequences (a pandas DataFrame, or the dict of
lists `read_tracker_csv` returns): pandas is optional here, the arithmetic is plain Python.

Deviation (documented): the reference's find_celeb_infor_in_interval indexes df['Emotion'] unconditionally and so
raises KeyError unless --recog_emotion was given; this build treats a missing Emotion column as "no emotions" (an
empty list per face).  Pinned by tests/golden/celeb_stat_ref.json, produced by running the reference's own
functions in the build container (tools/make_golden.py).
"""
import ast
import csv
import json
import math


def convert_sec_to_max_time_quantity(second):
    h = second // 3600
    remain_time = second % 3600
    m = remain_time // 60
    s = remain_time % 60
    return '{}h:{}m:{:.2f}s'.format(h, m, s)


def write_json(filename, content_dict, log=True):
    """utils/utils.py:40-45 (`indent=True` is json's indent=1)."""
    with open(filename, 'w') as fp:
        json.dump(content_dict, fp, indent=True)
    if log:
        print('Write json file {}'.format(filename))


def _col(df, name, default=None):
    try:
        return list(df[name])
    except (KeyError, IndexError):
        return default


def _rows(df, start, end):
    """df.iloc[start:end] for a DataFrame or a dict of lists."""
    if hasattr(df, "iloc"):
        return df.iloc[start:end]
    return {k: list(v)[start:end] for k, v in df.items()}


def find_celeb_infor_in_interval(df_for_itv, unknown_name, n_appear):
    names_c, bboxes_c, time_c = _col(df_for_itv, 'Names'), _col(df_for_itv, 'Bboxes'), _col(df_for_itv, 'Time')
    emo_c = _col(df_for_itv, 'Emotion')
    bboxes_dict = {}
    for r, (names_str, bboxes_str, time_s) in enumerate(zip(names_c, bboxes_c, time_c)):
        time_s = float(time_s)
        hms_time = convert_sec_to_max_time_quantity(time_s)
        list_names = ast.literal_eval(names_str)
        list_bboxes = ast.literal_eval(bboxes_str)
        list_emotions = ast.literal_eval(emo_c[r]) if emo_c is not None else [[] for _ in list_names]
        for name, bbox, emotion in zip(list_names, list_bboxes, list_emotions):
            bbox_item = {'time': hms_time, 'bbox': bbox, 'emotions': emotion}
            bboxes_dict.setdefault(name, []).append(bbox_item)
    final_bboxes_dict = {k: v for k, v in bboxes_dict.items() if k != unknown_name and len(v) >= n_appear}
    start_itv = convert_sec_to_max_time_quantity(float(time_c[0]))
    end_itv = convert_sec_to_max_time_quantity(float(time_c[-1]))
    return final_bboxes_dict, start_itv, end_itv


def _export(tracker_df, output_js_path, ranges, n_appear, unknown_name, log):
    dict_track = {}
    for i, (a, b) in enumerate(ranges):
        final_bboxes_dict, start_itv, end_itv = find_celeb_infor_in_interval(_rows(tracker_df, a, b), unknown_name, n_appear)
        dict_track[str(i + 1)] = {"interval": (start_itv, end_itv), "celebrities": final_bboxes_dict}
    write_json(output_js_path, dict_track, log=log)
    return dict_track


def export_json_stat_dynamic_itv(tracker_df, output_js_path, n_intervals, n_appear=4, unknown_name='Unknown', log=True):
    n_rows = len(_col(tracker_df, 'Time'))
    n_rows_in_itv = n_rows // n_intervals
    remain_rows = n_rows % n_intervals
    ranges = []
    for i in range(n_intervals):
        end_range = (i + 1) * n_rows_in_itv + (remain_rows if i == n_intervals - 1 else 0)
        ranges.append((i * n_rows_in_itv, end_range))
    return _export(tracker_df, output_js_path, ranges, n_appear, unknown_name, log)


def export_json_stat_fixed_itv(tracker_df, output_js_path, n_rows_in_itv, n_appear=4, unknown_name='Unknown', log=True):
    n_rows = len(_col(tracker_df, 'Time'))
    n_intervals = math.ceil(n_rows / n_rows_in_itv)
    ranges = [(i * n_rows_in_itv, min((i + 1) * n_rows_in_itv, n_rows)) for i in range(n_intervals)]
    return _export(tracker_df, output_js_path, ranges, n_appear, unknown_name, log)


def frame_is_sampled(count, fps, frame_idxes):
    """celeb_statistic.py:180-187: frame number `count` (1-based) is processed when count % fps is one of -fidx."""
    return any(count % fps == idx for idx in frame_idxes)


def build_thresholds(local_thresholds, num_classes, recog_threshold, read_json=None):
    """celeb_statistic.py:127-136: dict str(class) -> threshold."""
    if local_thresholds != '':
        if read_json is None:
            with open(local_thresholds) as fp:
                return json.load(fp)
        return read_json(local_thresholds)
    return {str(i): recog_threshold for i in range(num_classes)}


def tracker_header(track_bbox, recog_emotion=False):
    cols = ['Time', 'Names', 'Frame_idx']
    if track_bbox:
        cols.append('Bboxes')
    if recog_emotion:
        cols.append('Emotion')
    return cols


def tracker_row(time_in_video, names, frame_count, bboxes=None, frame_hw=None, track_bbox=False, emotions=None):
    """One tracker.csv line (celeb_statistic.py:253-276): boxes scaled to [0,1] by (w,h,w,h)."""
    row = [str(time_in_video), '"' + str(list(names)) + '"', str(frame_count)]
    if track_bbox and bboxes is not None:
        h, w = frame_hw
        scaled = [[float(b[0]) / w, float(b[1]) / h, float(b[2]) / w, float(b[3]) / h] for b in bboxes]
        row.append('"' + str(scaled) + '"')
    if emotions is not None:
        row.append('"' + str(emotions) + '"')
    return ','.join(row) + '\n'


def read_tracker_csv(path):
    """tracker.csv -> dict of column lists (what pd.read_csv gives the statistics functions, without pandas)."""
    with open(path, newline='') as fp:
        rd = csv.reader(fp)
        header = next(rd)
        cols = {h: [] for h in header}
        for rec in rd:
            if not rec:
                continue
            for h, v in zip(header, rec):
                cols[h].append(v)
    return cols
