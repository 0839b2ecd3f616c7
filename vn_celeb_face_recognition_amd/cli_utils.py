"""Small host helpers the CLIs share (formats kept from the reference):
  read_json, append_log_to_file, convert_sec_to_max_time_quantity <- /root/reference/utils/utils.py:34-38,60-64,77-82
  label2name CSV ('label,name' header)                             <- demo_image.py:359, meta_data/face_recognition/label2name.txt
  draw_boxes_on_image                                              <- demo_image.py:150-158 (PIL instead of OpenCV, which is not installed)
"""
import csv
import json

import numpy as np


def read_json(filename):
    with open(filename, 'r') as fp:
        return json.load(fp)


def append_log_to_file(file_path, list_items):
    with open(file_path, 'a') as f:
        f.write(','.join(list_items) + '\n')


def convert_sec_to_max_time_quantity(second):
    h = second // 3600
    remain_time = second % 3600
    m = remain_time // 60
    s = remain_time % 60
    return '{}h:{}m:{:.2f}s'.format(h, m, s)


def read_label2name(path):
    """CSV with header label,name -> {'label': [...], 'name': [...]} (column access like the DataFrame)."""
    labels, names = [], []
    with open(path, newline='') as f:
        for row in csv.DictReader(f):
            labels.append(int(row['label']))
            names.append(row['name'])
    return {'label': labels, 'name': names}


def read_rgb(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert('RGB'))


def write_rgb(path, rgb):
    from PIL import Image
    Image.fromarray(np.asarray(rgb, dtype=np.uint8)).save(path)


def draw_boxes_on_image(rgb_image, boxes, list_names):
    """Green 2-px rectangles with the name at the top-right corner (demo_image.py:150-158)."""
    from PIL import Image, ImageDraw
    im = Image.fromarray(np.asarray(rgb_image, dtype=np.uint8).copy())
    d = ImageDraw.Draw(im)
    for box, name in zip(boxes, list_names):
        d.rectangle([float(box[0]), float(box[1]), float(box[2]), float(box[3])], outline=(0, 255, 0), width=2)
        d.text((float(box[2]), float(box[1])), str(name), fill=(0, 255, 0))
    return np.asarray(im)


def export_video_face_recognition(output_frame_dir, fps, output_path):
    """demo_video.py:25-43: frame_1.png .. frame_N.png of the output directory -> one video.  cv2.VideoWriter (MP4V) when
    OpenCV is importable; otherwise a Motion-JPEG AVI (output_path must end in .avi)."""
    import glob
    import os
    n_images = len(glob.glob(os.path.join(output_frame_dir, '*')))
    paths = [os.path.join(output_frame_dir, 'frame_{}.png'.format(i)) for i in range(1, n_images + 1)]
    paths = [p for p in paths if os.path.exists(p)]
    if not paths:
        raise RuntimeError("export_video_face_recognition: no frame_<i>.png under %r (run with -sfr)" % output_frame_dir)
    try:
        import cv2
    except ImportError:
        cv2 = None
    if cv2 is not None:
        first = cv2.imread(paths[0])
        out_writer = cv2.VideoWriter(output_path, cv2.VideoWriter_fourcc(*'MP4V'), fps, (first.shape[1], first.shape[0]))
        for pth in paths:
            out_writer.write(cv2.imread(pth))
        out_writer.release()
    else:
        if not output_path.lower().endswith('.avi'):
            raise RuntimeError("without OpenCV the exported video is a Motion-JPEG AVI: give -ov a name ending in .avi")
        from .mjpeg_avi import write_mjpeg_avi
        write_mjpeg_avi(output_path, (read_rgb(pth) for pth in paths), fps)
    print('Save exported video in {} ...'.format(output_path))


def open_frame_source(path):
    """video.FrameSource over RGB frames.  Accepts a directory of images (sorted; random access: a rank decodes only its
    own batches), a .npy/.npz array of (T,H,W,3) uint8 frames (memory-mapped), a Motion-JPEG .avi (pure-Python RIFF
    walker, mjpeg_avi.py), or any video file when OpenCV is importable (demo_video.py:78-81; sequential: the other
    ranks' frames are decoded and dropped)."""
    import os
    from .video import FrameSource
    if os.path.isdir(path):
        files = sorted(f for f in os.listdir(path) if f.lower().endswith(('.png', '.jpg', '.jpeg', '.bmp')))
        return FrameSource([os.path.join(path, f) for f in files], 25.0, load=read_rgb)
    if path.endswith('.npy'):
        return FrameSource(np.load(path, mmap_mode='r'), 30.0)
    if path.endswith('.npz'):
        return FrameSource(np.load(path)['arr_0'], 30.0)
    try:
        import cv2
    except ImportError:
        cv2 = None
    if cv2 is None:
        from .mjpeg_avi import read_mjpeg_avi
        try:
            fps, frames, _ = read_mjpeg_avi(path)
        except (ValueError, OSError) as e:
            raise RuntimeError("decoding %r needs OpenCV, which is not installed (%s): pass a Motion-JPEG .avi, a directory "
                               "of frames or a .npy array of (T,H,W,3) uint8 RGB frames instead" % (path, e))
        return FrameSource(frames, fps)
    cap = cv2.VideoCapture(path)
    fps = cap.get(cv2.CAP_PROP_FPS) or 25.0

    def gen():
        while cap.isOpened():
            ret, frame = cap.read()
            if not ret:
                break
            yield frame[:, :, ::-1]
        cap.release()
    return FrameSource(gen(), fps)
