"""Decathlon-style data lists with the modality injected into every item -- drop-in for reference data/utils.py:11-60
(`load_decathlon_datalist_with_modality`), including MONAI's `_append_paths` (restated: MONAI 1.1.0 is not vendored).

The reference maps the JSON's `modality` entry (e.g. {"0": "CT"} / "CT" / "MR") through its data module into the integer style id the
conditional norms consume; `modality_id` below is that mapping made explicit (CT = 0, MR / MRI = 1: SURVEY.md section 0)."""
import json
import os
from pathlib import Path
from typing import Dict, List, Optional

MODALITY_IDS = {"ct": 0, "mr": 1, "mri": 1}


def _append_paths(base_dir, is_segmentation, items):
    """MONAI decathlon_datalist._append_paths: relative `image` (str or list of str) and, for segmentation lists, `label` paths are joined to
    base_dir; other keys are left alone.  TypeError for a non-dict item, like MONAI."""
    def fix(v):
        if isinstance(v, list):
            return [fix(i) for i in v]
        if isinstance(v, (str, os.PathLike)) and not os.path.isabs(v):
            return os.path.normpath(os.path.join(base_dir, v))
        return v
    for item in items:
        if not isinstance(item, dict):
            raise TypeError(f"Every item in items must be a dict but got {type(item).__name__}.")
        for k in ("image",) + (("label",) if is_segmentation else ()):
            if k in item:
                item[k] = fix(item[k])
    return items


def load_decathlon_datalist_with_modality(data_list_file_path, is_segmentation: bool = True, data_list_key: str = "training",
                                          base_dir: Optional[str] = None) -> List[Dict]:
    """same contract and error messages as reference data/utils.py:11-60"""
    data_list_file_path = Path(data_list_file_path)
    if not data_list_file_path.is_file():
        raise ValueError(f"Data list file {data_list_file_path} does not exist.")
    with open(data_list_file_path) as json_file:
        json_data = json.load(json_file)
    if data_list_key not in json_data:
        raise ValueError(f'Data list {data_list_key} not specified in "{data_list_file_path}".')
    expected_data = json_data[data_list_key]
    for data in expected_data:                      # reference :47-49 (items must be dicts here, as in the reference)
        data["modality"] = json_data["modality"]
    if data_list_key == "test" and not isinstance(expected_data[0], dict):
        expected_data = [{"image": i} for i in expected_data]
    if base_dir is None:
        base_dir = data_list_file_path.parent
    return _append_paths(base_dir, is_segmentation, expected_data)


def modality_id(modality) -> int:
    """"CT" / "MR" / "MRI" (any case), {"0": "CT"}-style decathlon dicts with one entry, or an int already"""
    if isinstance(modality, dict):
        if len(modality) != 1:
            raise ValueError(f"expected one modality per data list, got {modality}")
        modality = next(iter(modality.values()))
    if isinstance(modality, int):
        return modality
    key = str(modality).strip().lower()
    if key not in MODALITY_IDS:
        raise ValueError(f"unknown modality '{modality}' (known: {sorted(MODALITY_IDS)})")
    return MODALITY_IDS[key]
