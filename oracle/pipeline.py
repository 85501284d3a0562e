"""Reference-shaped CPU pipeline built from the oracle stages (TEST INFRASTRUCTURE ONLY).

  preprocess            TextDetector.transform + cvtColor   app/ml/models/text_detector.py:99-104,119-124
  detect                TextDetector.detect                 text_detector.py:115-141
  build_vocab           TextRecognizer._build_vocab         app/ml/models/text_recognizer.py:86-91
  decode_prediction     TextRecognizer._decode_prediction   text_recognizer.py:142-167
  recognize_batch       TextRecognizer._recognize_crnn_batch text_recognizer.py:114-140
  process_single_frame  VideoTextPipeline.process_single_frame   app/ml/inference/pipeliine.py:143-172
  process_frame_batch   VideoTextPipeline._process_frame_batch   pipeliine.py:93-141
  generate_summary      VideoTextPipeline._generate_summary      pipeliine.py:174-210
"""
import numpy as np
import torch

from . import cstages, nets

MEAN = np.array([0.485, 0.456, 0.406], np.float32)
STD = np.array([0.229, 0.224, 0.225], np.float32)


def preprocess(frame_bgr):
    """uint8 HWC BGR -> float32 [1,3,640,640] (RGB, Pillow bilinear 640x640, /255, normalise)."""
    rgb = frame_bgr[..., ::-1]
    small = cstages.pil_resize_bilinear(np.ascontiguousarray(rgb), 640, 640)
    x = torch.from_numpy(small).permute(2, 0, 1).float().div(255.0)  # ToTensor
    x = (x - torch.from_numpy(MEAN)[:, None, None]) / torch.from_numpy(STD)[:, None, None]
    return x.unsqueeze(0)


def detect(frame_bgr, sd, backbone, threshold=0.5, return_map=False):
    x = preprocess(frame_bgr)
    prob = nets.dbnet_forward(x, sd, backbone)["probability"][0, 0].numpy()
    dets = cstages.postprocess(prob, frame_bgr.shape[1], frame_bgr.shape[0], threshold)
    return (dets, prob) if return_map else dets


def build_vocab():
    chars = ("0123456789abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ"
             "!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~ ")
    vocab = {ch: i + 1 for i, ch in enumerate(chars)}
    vocab["<blank>"] = 0
    vocab["<unk>"] = len(vocab)
    return vocab


def decode_prediction(prob, vocab=None):
    """Greedy CTC with the reference's quirks: a blank does not reset ``prev``; '<unk>' emits nothing
    but does reset it; the confidence row is indexed by output length, not by timestep."""
    vocab = vocab or build_vocab()
    rev = {v: k for k, v in vocab.items()}
    prob = np.asarray(prob)
    idx = prob.argmax(axis=1)
    text, confs, prev = "", [], None
    for k in idx.tolist():
        if k == 0 or k == prev:
            continue
        ch = rev.get(k, "<unk>")
        if ch != "<unk>":
            text += ch
            confs.append(float(prob[len(text) - 1].max()))
        prev = k
    return text, (float(np.mean(confs)) if confs else 0.0)


def crop_tensor(crop_bgr):
    small = cstages.cv_resize_linear(crop_bgr, 128, 32)
    return torch.from_numpy(small).permute(2, 0, 1).float() / 255.0


def recognize_batch(crops, sd, vocab=None, return_probs=False):
    x = torch.stack([crop_tensor(c) for c in crops])
    probs = torch.softmax(nets.crnn_forward(x, sd), dim=2).numpy()
    out = []
    for p in probs:
        t, c = decode_prediction(p, vocab)
        out.append({"text": t, "confidence": c})
    return (out, probs) if return_probs else out


def process_single_frame(frame, det_sd, backbone, rec_sd, threshold=0.5):
    dets = detect(frame, det_sd, backbone, threshold)
    regions = []
    for d in dets:
        x1, y1, x2, y2 = d["bbox"]
        crop = frame[y1:y2, x1:x2]
        if crop.size == 0:
            continue
        r = recognize_batch([crop], rec_sd)[0]
        regions.append({"bbox": d["bbox"], "text": r["text"], "detection_confidence": d["confidence"],
                        "recognition_confidence": r["confidence"]})
    return {"detections": regions}


def process_frame_batch(frames, frame_info, det_sd, backbone, rec_sd, threshold=0.5):
    results = []
    for frame, (num, ts) in zip(frames, frame_info):
        dets = detect(frame, det_sd, backbone, threshold)
        regions = []
        for d in dets:
            x1, y1, x2, y2 = d["bbox"]
            crop = frame[y1:y2, x1:x2]
            if crop.size == 0:
                continue
            r = recognize_batch([crop], rec_sd)[0]
            regions.append({"bbox": d["bbox"], "text": r["text"], "detection_confidence": d["confidence"],
                            "recognition_confidence": r["confidence"], "polygon": d.get("polygon", [])})
        results.append({"frame_number": num, "timestamp": ts, "detections": regions})
    return results


def generate_summary(results, processing_time, frame_count):
    dets = [d for fr in results for d in fr["detections"]]
    texts = {d["text"].strip() for d in dets if d["text"].strip()}
    return {
        "total_frames": frame_count,
        "frames_with_text": sum(1 for fr in results if fr["detections"]),
        "total_detections": len(dets),
        "unique_texts": len(texts),
        "detected_texts": list(texts),
        "avg_detection_confidence": float(np.mean([d["detection_confidence"] for d in dets])) if dets else 0.0,
        "avg_recognition_confidence": float(np.mean([d["recognition_confidence"] for d in dets])) if dets else 0.0,
        "processing_time_seconds": processing_time,
        "fps_processed": frame_count / processing_time if processing_time > 0 else 0,
    }
