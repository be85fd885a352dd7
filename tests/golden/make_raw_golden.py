"""Writes tests/golden/raw_mlperf_val.bin with the REFERENCE's own converter for the Raw format of
the MLPerf DLRM-DCNv2 sample (R/samples/dlrm/preprocessing/convert_to_raw.py: DataConverter, stage
"val") from a small synthetic TorchRec-style input (int32 labels, float32 dense features, 26 int32
multi-hot arrays with the sample's hotness), and keeps the input arrays next to it
(raw_mlperf_inputs.npz).  tests/test_api_cpu.py reads the file with hugectr_amd.data.RawReader and
must get the inputs back.  Run in the build container:  python tests/golden/make_raw_golden.py"""
import importlib.util
import logging
import os
import shutil
import tempfile

import numpy as np

REF = "/root/reference/samples/dlrm/preprocessing/convert_to_raw.py"
HERE = os.path.dirname(os.path.abspath(__file__))
HOT = [3, 2, 1, 2, 6, 1, 1, 1, 1, 7, 3, 8, 1, 6, 9, 5, 1, 1, 1, 12, 100, 27, 10, 3, 1, 1]  # train.py:58-85
N = 96


def main():
    spec = importlib.util.spec_from_file_location("convert_to_raw_ref", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    rng = np.random.default_rng(3)
    labels = rng.integers(0, 2, size=N).astype(np.int32)
    dense = rng.standard_normal((N, 13)).astype(np.float32)
    sparse = {str(i): rng.integers(0, 1 << 30, size=(N, h)).astype(np.int32) for i, h in enumerate(HOT)}
    tmp = tempfile.mkdtemp()
    try:
        day = ref.NUM_DAYS - 1
        np.save(os.path.join(tmp, ref.INPUT_LABELS_FILE.format(day=day)), labels)
        np.save(os.path.join(tmp, ref.INPUT_DENSE_FILE.format(day=day)), dense)
        np.savez(os.path.join(tmp, ref.INPUT_SPARSE_FILE.format(day=day)), **sparse)
        conv = ref.DataConverter(tmp, tmp, tmp, ref.VAL, 1 << 20, 40, logging.getLogger("raw"), 10 ** 9)
        conv.save()
        shutil.copy(os.path.join(tmp, ref.OUTPUT_FILE.format(stage=ref.VAL)),
                    os.path.join(HERE, "raw_mlperf_val.bin"))
    finally:
        shutil.rmtree(tmp)
    np.savez_compressed(os.path.join(HERE, "raw_mlperf_inputs.npz"), labels=labels, dense=dense, **sparse)
    print("wrote", os.path.getsize(os.path.join(HERE, "raw_mlperf_val.bin")), "bytes")


if __name__ == "__main__":
    main()
