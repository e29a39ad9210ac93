"""Training entry point: JSON configuration -> model, optimizer, checkpoints, epoch loop
(mirror of the reference's athena/main.py:37-198 for the models this package builds).

    python -m athena_amd.main config.json

The top-level keys and their defaults are the reference's (main.py:58-82); unknown keys raise
``KeyError`` as there.  ``dataset_builder`` names come from ``SUPPORTED_DATASET_BUILDER``: this
package ships the synthetic builder only, the corpus-reading builders of the reference are named so
that a configuration using them fails with a clear message instead of a missing key.
"""
import json
import logging
import sys

import torch

from .data import SyntheticSpeechRecognitionDatasetBuilder
from .models.mtl_seq2seq import MtlTransformerCtc
from .models.speech_transformer import SpeechTransformer
from .solver import BaseSolver
from .utils.checkpoint import Checkpoint
from .utils.hparam import register_and_parse_hparams
from .utils.learning_rate import ExponentialDecayAdam, WarmUpAdam
from .utils.metric_check import set_default_summary_writer


def _not_built(name):
    def build(*args, **kwargs):
        raise NotImplementedError("%r reads real corpora (csv, vocabulary, cmvn) and is not part of this package; "
                                  "use 'synthetic_speech_recognition' or pass a builder object" % name)
    return build


SUPPORTED_DATASET_BUILDER = {
    "synthetic_speech_recognition": SyntheticSpeechRecognitionDatasetBuilder,
    "speech_recognition_dataset": _not_built("speech_recognition_dataset"),
    "speech_recognition_dataset_kaldiio": _not_built("speech_recognition_dataset_kaldiio"),
}

SUPPORTED_MODEL = {
    "speech_transformer": SpeechTransformer,
    "mtl_transformer_ctc": MtlTransformerCtc,
}

SUPPORTED_OPTIMIZER = {
    "warmup_adam": WarmUpAdam,
    "expdecay_adam": ExponentialDecayAdam,
}

DEFAULT_CONFIGS = {
    "batch_size": 32,
    "num_epochs": 20,
    "sorta_epoch": 1,
    "ckpt": None,
    "summary_dir": None,
    "solver_type": "asr",
    "solver_gpu": [0],
    "solver_config": None,
    "model": "speech_transformer",
    "num_classes": None,
    "model_config": None,
    "pretrained_model": None,
    "teacher_model": None,
    "optimizer": "warmup_adam",
    "optimizer_config": None,
    "convert_config": None,
    "num_data_threads": 1,
    "dataset_builder": "speech_recognition_dataset",
    "dev_dataset_builder": None,
    "trainset_config": None,
    "devset_config": None,
    "testset_config": None,
    "inference_config": None
}


def parse_config(config):
    """dict -> hparams (main.py:84-88)"""
    p = register_and_parse_hparams(DEFAULT_CONFIGS, config, cls="main")
    logging.info("hparams: %s", p)
    return p


def parse_jsonfile(jsonfile):
    """json file -> hparams (main.py:90-98)"""
    with open(jsonfile) as file:
        config = json.load(file)
    return parse_config(config)


def build_model_from_jsonfile(jsonfile, pre_run=True):
    """Model, optimizer and checkpoint manager of a configuration; restores the latest snapshot of
    the checkpoint directory if there is one (main.py:100-128)."""
    p = parse_jsonfile(jsonfile)
    if p.model not in SUPPORTED_MODEL:
        raise NotImplementedError("model %r is not built by this package (built: %s)" %
                                  (p.model, ", ".join(sorted(SUPPORTED_MODEL))))
    dataset_builder = SUPPORTED_DATASET_BUILDER[p.dataset_builder](p.trainset_config)
    if p.trainset_config is None and p.num_classes is None:
        raise ValueError("trainset_config and num_classes can not both be null")
    model = SUPPORTED_MODEL[p.model](data_descriptions=dataset_builder, config=p.model_config)
    optimizer = SUPPORTED_OPTIMIZER[p.optimizer](p.optimizer_config)
    optimizer.attach(model.store)
    checkpointer = Checkpoint(checkpoint_directory=p.ckpt, model=model, optimizer=optimizer)
    if pre_run or p.pretrained_model is not None:
        # the reference runs one evaluation step here to make Keras create the variables; here the
        # variables exist already and the step only checks that the configuration runs
        solver = BaseSolver(model, optimizer, sample_signature=dataset_builder.sample_signature)
        dataset = dataset_builder.as_dataset(p.batch_size)
        solver.evaluate_step(model.prepare_samples(next(iter(dataset))))
        model.reset_metrics()
    return p, model, optimizer, checkpointer


def train(jsonfile, Solver, rank_size=1, rank=0):
    """The epoch loop: train, evaluate, checkpoint (main.py:131-185)."""
    p, model, optimizer, checkpointer = build_model_from_jsonfile(jsonfile)
    epoch = int(checkpointer.save_counter)
    if p.pretrained_model is not None and epoch == 0:
        p2, pretrained_model, _, _ = build_model_from_jsonfile(p.pretrained_model)
        model.restore_from_pretrained_model(pretrained_model, p2.model)
    if p.teacher_model is not None:
        raise NotImplementedError("teacher_model (knowledge distillation of the TTS models) is not built")
    if rank == 0:
        set_default_summary_writer(p.summary_dir)

    trainset_builder = SUPPORTED_DATASET_BUILDER[p.dataset_builder](p.trainset_config)
    trainset_builder.compute_cmvn_if_necessary(rank == 0)
    trainset_builder.shard(rank_size, rank)

    devset_builder = p.dev_dataset_builder if p.dev_dataset_builder is not None else p.dataset_builder
    devset_builder = SUPPORTED_DATASET_BUILDER[devset_builder](p.devset_config)

    solver = Solver(model, optimizer, sample_signature=trainset_builder.sample_signature,
                    eval_sample_signature=devset_builder.sample_signature, config=p.solver_config)
    loss, metrics = 0.0, {}
    while epoch < p.num_epochs:
        if rank == 0:
            logging.info(">>>>> start training in epoch %d", epoch)
        if epoch >= p.sorta_epoch:
            trainset_builder.batch_wise_shuffle(p.batch_size)
        solver.train(trainset_builder.as_dataset(p.batch_size, p.num_data_threads))

        if rank == 0:
            logging.info(">>>>> start evaluate in epoch %d", epoch)
        devset = devset_builder.as_dataset(p.batch_size, p.num_data_threads)
        loss, metrics = solver.evaluate(devset, epoch)

        if rank == 0:
            checkpointer(loss, metrics)
        epoch = epoch + 1
    return loss, metrics


def main(argv=None):
    argv = sys.argv if argv is None else argv
    logging.basicConfig(level=logging.INFO)
    if len(argv) < 2:
        logging.warning("Usage: python %s config_json_file", argv[0])
        return 1
    torch.manual_seed(1)                    # tf.random.set_seed(1) (main.py:193)
    p = parse_jsonfile(argv[1])
    BaseSolver.initialize_devices(p.solver_gpu)
    train(argv[1], BaseSolver, 1, 0)
    return 0


if __name__ == "__main__":
    sys.exit(main())
