"""The reference's own C++ test cases (src/tests.cpp), restated in tests/cpp/reader_tests.cpp against this
project's memb::Builder / memb::Reader / CompressionStrategy, compiled without Python in the loop."""
import os
import subprocess

import pytest

from conftest import REPO

SOURCES = ['reader.cpp', 'builder.cpp', 'compression_strategy.cpp']


@pytest.fixture(scope='module')
def reader_tests(native, tmp_path_factory):
    directory = tmp_path_factory.mktemp('cpp')
    binary = str(directory / 'reader_tests')
    library_dir = os.path.dirname(native.HIP_LIBRARY_PATH)
    command = ['g++', '-O2', '-std=c++17', '-Wall', '-Werror', '-I', os.path.join(REPO, 'include'),
               os.path.join(REPO, 'tests', 'cpp', 'reader_tests.cpp')]
    command += [os.path.join(REPO, 'memb_amd', 'csrc', name) for name in SOURCES]
    command += ['-L', library_dir, '-lmemb_hip', '-Wl,-rpath,' + library_dir, '-pthread', '-o', binary]
    build = subprocess.run(command, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert build.returncode == 0, build.stdout
    return binary, str(directory)


def test_refusals_of_the_cpp_interface(reader_tests):
    binary, directory = reader_tests
    run = subprocess.run([binary, '--host'], cwd=directory, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                         timeout=120)
    assert run.returncode == 0, run.stdout


@pytest.mark.gpu
def test_reference_cpp_cases(reader_tests):
    binary, directory = reader_tests
    run = subprocess.run([binary], cwd=directory, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert run.returncode == 0, run.stdout
    assert 'trained storage, first-level table of 1 bit' in run.stdout and run.stdout.strip().endswith('ok (0 failed checks)')
