"""Entry points with the reference's command lines: `python -m lwm_amd.cli.train`,
`python -m lwm_amd.cli.vision_chat`, `python -m lwm_amd.cli.vision_generation` accept the flag names of
lwm/train.py:31-56, lwm/vision_chat.py:22-37 and lwm/vision_generation.py:21-41 (the scripts under the
reference's scripts/ run unchanged after `s/lwm\\./lwm_amd.cli./`).  They are thin: configuration ->
lwm_amd model harness -> the HIP hot path.  What the reference reaches through tux / optax / wandb / GCS
(datasets, loggers, checkpoint streaming to buckets) is accepted on the command line and reported as
unused; the data fed to the model is synthetic unless a local file is given."""
