FROM python:3.9-slim
RUN apt-get update && apt-get install -y --no-install-recommends git g++ make && rm -rf /var/lib/apt/lists/*
COPY . /work/build
RUN git clone --depth 1 https://github.com/tud-amr/multi-robot-fabrics /work/reference
RUN pip install --require-hashes --no-deps -r /work/build/tests/golden/reference_requirements.txt
WORKDIR /work/build
CMD python tests/golden/make_reference_golden.py --reference /work/reference \
 && python -m pytest tests/test_reference_pin.py -q -m "not gpu" ; \
    python tests/reconcile_constants.py --write multi-robot-fabrics_amd/constants.json
