import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes a minute or more (deselect with -m 'gpu and not slow')")


@pytest.fixture(scope="session")
def lubm1():
    """One LUBM-shaped university (degrees drawn from 3 universities so that Q2 has answers),
    its class index for 4 partitions and the oracle's artefacts for it."""
    from opus_amd import lubm
    from oracle import dap_oracle
    arr, n_triples = lubm.generate(0, 1, seed=0, ontology=0, degree_pool=3)
    text = arr.tobytes().decode()
    ci = lubm.class_index_text(4, 0)
    ref = dap_oracle.run_dap(text, ci, 4)
    return {"bytes": arr, "text": text, "n_triples": n_triples, "class_index": ci, "n_partitions": 4, "oracle": ref}


@pytest.fixture(scope="session")
def lubm1_swat():
    """The same university in the namespace of the reference's queries/lubm10240 folder (swat ontology IRIs)."""
    from opus_amd import lubm
    from oracle import dap_oracle
    arr, n_triples = lubm.generate(0, 1, seed=0, ontology=1, degree_pool=3)
    text = arr.tobytes().decode()
    ci = lubm.class_index_text(4, 1)
    ref = dap_oracle.run_dap(text, ci, 4)
    return {"bytes": arr, "text": text, "n_triples": n_triples, "class_index": ci, "n_partitions": 4, "oracle": ref}
