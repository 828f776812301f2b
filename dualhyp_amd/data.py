"""Prompt packing for GER / DualHyp / RelPrompt (the step in front of the decoder).

Restates the text side of the reference's datasets — templates `data/prompts.py:3-19`,
`AVDataset.get_prompt` `data/av_dataset.py:210-256`, `DualHypothesesAVDataset.get_prompt` `:373-429`,
`DualHypothesesMaskAVDataset` reliability tokens `:447-500,546-605`, `collate_fn` `:258-292` — without
the audio/video loading (`:151-208`), which the LLM path never reads.  Works on the merged JSON
schema of `data/merge_json.py:5-63`.
"""
from __future__ import annotations

import json
import random
from collections import OrderedDict
from typing import Any, Dict, List, Optional, Sequence, Tuple

import torch

PROMPTS: Dict[str, Dict[str, str]] = {
    "GER": {
        "prompt_1": "Below is the best-hypotheses transcribed from speech recognition system. Please try to revise it "
                    "using the words which are only included into other-hypothesis, and write the response for the "
                    "true transcription.\n\n### Best-hypothesis:\n",
        "prompt_2": "\n\n### Other-hypothesis:",
        "prompt_3": "\n\n### Response:\n",
    },
    "DualHyp": {
        "prompt_1": "Below are the best-hypothesis transcribed from speech recognition systems, ASR and VSR, "
                    "respectively. Please try to revise it using the words which are only included into "
                    "other-hypotheses, and write the response for the true transcription.\n\n### ASR Best-hypothesis:\n"
                    "<<<ASR_NHYPS>>>\n\n### VSR Best-hypothesis:\n<<<VSR_NHYPS>>>",
        "prompt_2": "\n\n### ASR Other-hypotheses:\n<<<ASR_NHYPS>>>\n\n### VSR Other-hypotheses:\n<<<VSR_NHYPS>>>",
        "prompt_3": "\n\n### Response:\n",
    },
    "RelPrompt": {
        "prompt_1": "Below are the best-hypothesis transcribed from speech recognition systems, ASR and VSR, "
                    "respectively. Please try to revise it using the words which are only included into "
                    "other-hypotheses, and write the response for the true transcription. Refer to the audio and video "
                    "masks for reliability.\n\n\n### ASR Best-hypothesis:\n<<<ASR_BEST_NHYPS>>>\n\n### ASR "
                    "Other-hypotheses:\n<<<ASR_NHYPS>>>\n\n### Audio Mask:\n<<<ASR_MASKS>>>\n\n\n### VSR Best-hypothesis:\n"
                    "<<<VSR_BEST_NHYPS>>>\n\n### VSR Other-hypotheses:\n<<<VSR_NHYPS>>>\n\n### Video Mask:\n<<<VSR_MASKS>>>",
        "prompt_2": "",
        "prompt_3": "\n\n\n### Response:\n",
    },
}
MASK_TOKENS = ("<<C>>", "<<M>>", "<<N>>")   # ids vocab..vocab+2 (inference/relprompt.py:341)


def get_prompts_format(name: str) -> Dict[str, str]:
    if name not in PROMPTS:
        raise ValueError(f"Unknown prompt name: {name}")
    return PROMPTS[name]


def _others(hyps: Sequence[str], max_nhyps: Optional[int]) -> List[str]:
    # random_sample_sequence(lst, len(lst)) in the reference is an order-preserving identity
    # (data/utils.py:250-255), so "random sampling" of all hypotheses is omitted here.
    return list(hyps[1:max_nhyps] if max_nhyps is not None else hyps[1:])


def ger_prompt(sample: Dict[str, Any], nhyps_key: str = "nhyps_asr", max_nhyps: Optional[int] = None) -> str:
    """data/av_dataset.py:224 — note the single newline glued in front of the other hypotheses."""
    p = PROMPTS["GER"]
    hyps = sample[nhyps_key]["hyps"]
    return p["prompt_1"] + hyps[0] + p["prompt_2"] + "\n" + "\n".join(_others(hyps, max_nhyps)) + p["prompt_3"]


def dualhyp_prompt(sample_asr: Dict[str, Any], sample_vsr: Dict[str, Any], max_nhyps: Optional[int] = None,
                   language: Optional[str] = None) -> str:
    """data/av_dataset.py:373-399."""
    p = PROMPTS["DualHyp"]
    p1 = p["prompt_1"]
    if language is not None:   # data/av_dataset.py:341-342 (no-op on this template: the phrase is plural there)
        p1 = p1.replace("speech recognition system", f"{language} speech recognition system")
    a, v = sample_asr["nhyps_asr"]["hyps"], sample_vsr["nhyps_vsr"]["hyps"]
    return (p1.replace("<<<ASR_NHYPS>>>", a[0]).replace("<<<VSR_NHYPS>>>", v[0])
            + p["prompt_2"].replace("<<<ASR_NHYPS>>>", "\n".join(_others(a, max_nhyps)))
                           .replace("<<<VSR_NHYPS>>>", "\n".join(_others(v, max_nhyps)))
            + p["prompt_3"])


def noise_mask(sample: Dict[str, Any], modality: str, mask_threshold: Optional[float] = None) -> List[str]:
    """Per-frame 'C'/'N' labels from the corruption record (data/av_dataset.py:447-473)."""
    if modality == "audio":
        c, snr = sample["Audio_Corruption"], sample["Audio_Corruption"]["snr"]
    elif modality == "video":
        c, snr = sample["Visual_Corruption"], -100
    else:
        raise ValueError("Invalid modality. Choose 'audio' or 'video'.")
    mask = ["C"] * c["total_len"]
    if mask_threshold is None or snr < mask_threshold:
        mask[c["start_fr"]:c["start_fr"] + c["occ_len"]] = ["N"] * c["occ_len"]
    return mask


def chunk_reliability(mask: Sequence[str], chunk_size: int, prefix: str = "") -> Tuple[List[float], List[str]]:
    """Clean fraction per chunk and its token: > 0.9 clean, < 0.6 noisy, else mixed (data/av_dataset.py:475-500)."""
    scores, labels = [], []
    for i in range(0, len(mask), chunk_size):
        chunk = mask[i:i + chunk_size]
        s = chunk.count("C") / len(chunk)
        scores.append(s)
        labels.append(f"<<{prefix}C>>" if s > 0.9 else f"<<{prefix}N>>" if s < 0.6 else f"<<{prefix}M>>")
    return scores, labels


def relprompt_prompt(sample_asr: Dict[str, Any], sample_vsr: Dict[str, Any], audio_labels: Sequence[str],
                     video_labels: Sequence[str], max_nhyps: Optional[int] = None, leave_masks: bool = False) -> str:
    """data/av_dataset.py:546-571."""
    p = PROMPTS["RelPrompt"]
    a, v = sample_asr["nhyps_asr"]["hyps"], sample_vsr["nhyps_vsr"]["hyps"]
    s = (p["prompt_1"].replace("<<<ASR_BEST_NHYPS>>>", a[0]).replace("<<<VSR_BEST_NHYPS>>>", v[0])
         .replace("<<<ASR_NHYPS>>>", "\n".join(_others(a, max_nhyps)))
         .replace("<<<VSR_NHYPS>>>", "\n".join(_others(v, max_nhyps))))
    if not leave_masks:
        s = s.replace("<<<ASR_MASKS>>>", "".join(audio_labels)).replace("<<<VSR_MASKS>>>", "".join(video_labels))
    return s + p["prompt_3"]


def encode_example(tokenizer, prompt_no_response: str, caption: str, max_input_length: int = 0) -> Dict[str, Any]:
    """Token ids + labels (-1 over the prompt) — data/av_dataset.py:246-256, 360-362.
    `tokenizer` needs `.encode(str) -> List[int]` and `.eos_token`."""
    full = prompt_no_response + caption + tokenizer.eos_token
    ids_np = tokenizer.encode(prompt_no_response)
    ids = tokenizer.encode(full)
    labels = [-1] * len(ids_np) + ids[len(ids_np):]
    ids_t, lab_t = torch.tensor(ids, dtype=torch.int64), torch.tensor(labels, dtype=torch.int64)
    if max_input_length > 0:
        ids_t, lab_t = ids_t[:max_input_length], lab_t[:max_input_length]
    return {"input_ids": ids_t, "labels": lab_t, "input_ids_no_response": torch.tensor(ids_np, dtype=torch.int64),
            "input": full, "input_no_response": prompt_no_response}


def collate(samples: Sequence[Dict[str, Any]]) -> Dict[str, Any]:
    """Right-pad ids with 0 and labels with -1 (data/av_dataset.py:258-292, text part)."""
    n = max(s["input_ids"].size(0) for s in samples)

    def pad(t: torch.Tensor, v: int) -> torch.Tensor:
        return torch.cat([t, torch.full((n - t.size(0),), v, dtype=t.dtype)])
    out = {"input_ids": torch.stack([pad(s["input_ids"], 0) for s in samples]),
           "labels": torch.stack([pad(s["labels"], -1) for s in samples]),
           "input_ids_no_response": [s.get("input_ids_no_response") for s in samples],
           "input": [s.get("input", "") for s in samples],
           "uid": [s.get("uid", "") for s in samples],
           "ground_truth": [s.get("ground_truth", "") for s in samples]}
    # RelPrompt fine-tune (finetune/relprompt.py:346-364): encoder features [T, C] and per-chunk reliability class
    # indices ride along when an example carries them; like the reference's torch.stack they must agree in length
    for k in ("audio_enc_features", "visual_enc_features", "audio_mask_targets", "visual_mask_targets"):
        if all(k in s for s in samples):
            out[k] = torch.stack([s[k] for s in samples])
    return out


def feature_key(item: Dict[str, Any], stream: str, n_variants: int) -> str:
    """File stem of an utterance variant's encoder features (RelPrompt; `--enc_features_dir`): the Uid when the JSON holds ONE
    item for it, else Uid + a stable hash of that item's corruption record of `stream` ("Audio_Corruption" / "Visual_Corruption"),
    so features and mask targets always describe the same corruption."""
    if n_variants <= 1:
        return str(item["Uid"])
    import hashlib
    rec = json.dumps(item.get(stream, {}), sort_keys=True)
    return f"{item['Uid']}.{hashlib.sha1(rec.encode()).hexdigest()[:10]}"


class HypothesesDataset:
    """JSON -> examples.  Items sharing a `Uid` are alternative corruptions of one utterance; the dual
    formats draw the ASR and the VSR item independently with `random.choices(k=2)`
    (data/av_dataset.py:68-79, 343-346)."""

    def __init__(self, json_path_or_items, tokenizer, prompts_format: str = "DualHyp", nhyps_key: str = "nhyps_asr",
                 max_nhyps: Optional[int] = None, max_input_length: int = 0, language: Optional[str] = None,
                 mask_threshold: Optional[float] = None, time_window: float = 0.4, seed: Optional[int] = None,
                 enc_features=None, leave_masks: bool = False) -> None:
        items = json_path_or_items
        if isinstance(items, (str, bytes)) or hasattr(items, "__fspath__"):
            with open(items, encoding="utf-8") as f:
                items = json.load(f)
        self.uid2sample: "OrderedDict[str, List[Dict[str, Any]]]" = OrderedDict()
        for it in items:
            self.uid2sample.setdefault(it["Uid"], []).append(it)
        self.uids = list(self.uid2sample)
        # RelPrompt fine-tune: callable (asr_item, vsr_item) -> (audio encoder features [T_a, whisper_dim], visual
        # [T_v, raven_dim]); the Whisper / BRAVEn encoders themselves are upstream of this path (finetune/relprompt.py:346-352)
        self.enc_features = enc_features
        # RelPrompt inference (inference/relprompt.py:113-153): the prompt keeps its <<<ASR_MASKS>>> / <<<VSR_MASKS>>> placeholders;
        # the harness fills them with the classifiers' predictions and re-encodes
        self.leave_masks = leave_masks
        self.tokenizer, self.fmt, self.nhyps_key = tokenizer, prompts_format, nhyps_key
        self.max_nhyps, self.max_input_length, self.language = max_nhyps, max_input_length, language
        self.mask_threshold = mask_threshold
        self.audio_chunk, self.video_chunk = int(16000 * time_window), int(25 * time_window)
        self.rng = random.Random(seed) if seed is not None else random

    def __len__(self) -> int:
        return len(self.uids)

    def __getitem__(self, i: int) -> Dict[str, Any]:
        group = self.uid2sample[self.uids[i]]
        if self.fmt == "GER":
            s1 = s2 = self.rng.choice(group)
            prompt = ger_prompt(s1, self.nhyps_key, self.max_nhyps)
        else:
            s1, s2 = self.rng.choices(group, k=2)
            if self.fmt == "DualHyp":
                prompt = dualhyp_prompt(s1, s2, self.max_nhyps, self.language)
            else:
                _, al = chunk_reliability(noise_mask(s1, "audio", self.mask_threshold), self.audio_chunk)
                _, vl = chunk_reliability(noise_mask(s2, "video", self.mask_threshold), self.video_chunk)
                prompt = relprompt_prompt(s1, s2, al, vl, self.max_nhyps, leave_masks=self.leave_masks)
        ex = encode_example(self.tokenizer, prompt, s1["Caption"], self.max_input_length)
        ex["uid"], ex["ground_truth"] = s1.get("Uid", ""), s1.get("Caption", "")
        if self.fmt == "RelPrompt":
            # ground-truth reliability classes of the chunks (finetune/relprompt.py:73-79,364-365: <<C>> 0, <<M>> 1, else 2)
            cls = {MASK_TOKENS[0]: 0, MASK_TOKENS[1]: 1}
            ex["audio_mask_targets"] = torch.tensor([cls.get(l, 2) for l in al], dtype=torch.int64)
            ex["visual_mask_targets"] = torch.tensor([cls.get(l, 2) for l in vl], dtype=torch.int64)
            if self.enc_features is not None:
                # the features must describe the SAME corruption as the mask targets built from s1 / s2 above: a Uid with several
                # noise variants is served per variant (feature_key below), never by the Uid alone
                ex["audio_enc_features"], ex["visual_enc_features"] = self.enc_features(s1, s2)
        return ex
