FROM python:3.9-slim
RUN apt-get update && apt-get install -y --no-install-recommends git g++ make && rm -rf /var/lib/apt/lists/*
COPY . /work/build
# Not --depth 1 at whatever HEAD is: the environment below is hash-pinned to ONE poetry.lock and SURVEY.md cites file:line of
# one revision.  make_reference_golden.py refuses a checkout whose path files differ from the surveyed ones
# (EXPECTED_SHA256) and records git HEAD + the lock's content-hash in reference_provenance.npz; if upstream has moved,
# `git -C /work/reference checkout <commit>` the revision whose files match before running it (REFERENCE_COMMIT).
ARG REFERENCE_COMMIT=HEAD
RUN git clone https://github.com/tud-amr/multi-robot-fabrics /work/reference && git -C /work/reference checkout ${REFERENCE_COMMIT}
RUN pip install --require-hashes --no-deps -r /work/build/tests/golden/reference_requirements.txt
WORKDIR /work/build
CMD python tests/golden/make_reference_golden.py --reference /work/reference \
 && python -m pytest tests/test_reference_pin.py -q -m "not gpu" ; \
    python tests/reconcile_constants.py --write multi-robot-fabrics_amd/constants.json
