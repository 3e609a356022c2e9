"""CPU: the reference-side binding (bindings/jni/icp_jni.c + bindings/scala/api/gpu/*.scala) against the C ABI.

No JDK and no Scala compiler in the image, so:
  - every `@native def` of api.gpu.NativeIcp has a Java_api_gpu_NativeIcp_00024_<name> definition in icp_jni.c with the same number of
    arguments (and the other way round), and the JNI types match the Scala types position by position;
  - every icp_* function the shim calls is declared in include/icp_proposal.h, and the shim compiles with -Wall -Wextra -Werror against
    that header and the JNI test double of tests/support/jni_mock (argument types of every ABI call checked by the compiler);
  - no critical region is held anywhere (every native blocks on the GPU);
  - the Scala adapters only call natives that exist, with the declared number of arguments.
tests/test_gpu_jni.py RUNS the natives on the GPU box through the same test double."""
import os
import re
import subprocess

from conftest import ROOT

JNI_C = os.path.join(ROOT, "bindings", "jni", "icp_jni.c")
SCALA_DIR = os.path.join(ROOT, "bindings", "scala", "api", "gpu")
HEADER = os.path.join(ROOT, "include", "icp_proposal.h")

SCALA_TO_JNI = {"Int": "jint", "Long": "jlong", "Double": "jdouble", "Boolean": "jboolean", "Unit": "void",
                "Array[Double]": "jdoubleArray", "Array[Int]": "jintArray", "Array[Long]": "jlongArray"}


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def scala_natives():
    text = _strip_comments(open(os.path.join(SCALA_DIR, "NativeIcp.scala")).read())
    out = {}
    for m in re.finditer(r"@native\s+def\s+(\w+)\s*\((.*?)\)\s*:\s*([\w\[\]]+)", text, flags=re.S):
        args = [a.split(":")[1].strip() for a in m.group(2).split(",") if a.strip()]
        out[m.group(1)] = (args, m.group(3))
    return out


def jni_definitions():
    text = _strip_comments(open(JNI_C).read())
    out = {}
    for m in re.finditer(r"NATIVE\((\w+),\s*(\w+)\)\s*\((.*?)\)\s*\{", text, flags=re.S):
        args = [" ".join(a.replace("*", " ").split()[:-1]) for a in m.group(3).split(",")]
        assert args[0] == "JNIEnv" and args[1] == "jobject", (m.group(2), args[:2])
        out[m.group(2)] = (args[2:], m.group(1))
    return out


def test_every_scala_native_has_a_jni_definition_and_back():
    sc, jn = scala_natives(), jni_definitions()
    assert len(sc) >= 28
    assert set(sc) == set(jn), (sorted(set(sc) - set(jn)), sorted(set(jn) - set(sc)))
    for name, (args, ret) in sc.items():
        jargs, jret = jn[name]
        assert [SCALA_TO_JNI[a] for a in args] == jargs, (name, args, jargs)
        assert SCALA_TO_JNI[ret] == jret, (name, ret, jret)
    # the boundary of rounds 4-6 is bound, not only the three plug-in methods
    for must in ("ctxCreateKeyed", "ctxSetTarget", "chainBind", "chainStepBatched", "chainStepBatchedIssue", "chainStepBatchedCollect",
                 "chainStepBatchedAbandon", "chainsRunOnDevice", "propose", "logTransition", "logValue"):
        assert must in sc


def test_every_abi_call_of_the_shim_is_declared_in_the_header():
    declared = set(re.findall(r"ICP_API\s+[\w\s\*]+?\b(icp_\w+)\s*\(", open(HEADER).read()))
    called = set(re.findall(r"\b(icp_[a-z_0-9]+)\s*\(", _strip_comments(open(JNI_C).read())))
    assert called and called <= declared, sorted(called - declared)
    for must in ("icp_ctx_create_keyed", "icp_ctx_set_target", "icp_chain_bind", "icp_chain_step_batched", "icp_chain_step_batched_issue",
                 "icp_chain_step_batched_collect", "icp_chain_step_batched_abandon", "icp_chains_run_on_device", "icp_chain_step"):
        assert must in called, must


def test_shim_compiles_against_the_header_with_the_jni_test_double():
    """-fsyntax-only -Werror with tests/support/jni_mock/jni.h standing in for the JDK's: a wrong argument type or count in any ABI call
    of the shim is a compile error here.  (Without the double the translation unit is empty by construction — also checked.)"""
    mock = os.path.join(ROOT, "tests", "support", "jni_mock")
    base = ["gcc", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include")]
    done = subprocess.run(base + ["-I", mock, JNI_C], capture_output=True, text=True)
    assert done.returncode == 0, done.stderr
    done = subprocess.run(base + ["-Wno-unused", JNI_C], capture_output=True, text=True)
    assert done.returncode == 0, done.stderr
    sym = subprocess.run(["gcc", "-E", "-dM", "-I", mock, "-I", os.path.join(ROOT, "include"), JNI_C], capture_output=True, text=True).stdout
    assert "ICP_HAVE_JNI" in sym


def test_no_critical_region_and_mixture_struct_size():
    text = _strip_comments(open(JNI_C).read())
    assert "PrimitiveArrayCritical" not in text   # (verdict r05: a critical region held across a blocking GPU call stalls every GC)
    assert "mix.struct_size = sizeof(icp_mh_mixture)" in text
    for field in ("w_pose", "pose_rot_sigma", "pose_trans_sigma", "rw_sigma", "w_icp", "w_rw", "icp_weight"):
        assert "mix." + field in text, field


def test_scala_adapters_call_existing_natives_with_the_declared_arity():
    sc = scala_natives()
    seen = set()
    for fn in sorted(os.listdir(SCALA_DIR)):
        if fn == "NativeIcp.scala":
            continue
        text = _strip_comments(open(os.path.join(SCALA_DIR, fn)).read())
        for m in re.finditer(r"NativeIcp\.(\w+)\s*\(", text):
            name = m.group(1)
            assert name in sc, (fn, name)
            depth, i, n_args, any_arg = 1, m.end(), 1, False
            while depth:
                ch = text[i]
                if ch in "([":
                    depth += 1
                elif ch in ")]":
                    depth -= 1
                elif ch == "," and depth == 1:
                    n_args += 1
                if depth and not ch.isspace():
                    any_arg = True
                i += 1
            assert (n_args if any_arg else 0) == len(sc[name][0]), (fn, name, n_args, len(sc[name][0]))
            seen.add(name)
    # the adapters use the new boundary
    for must in ("ctxCreateKeyed", "ctxSetTarget", "chainBind", "chainsRunOnDevice", "propose", "logTransition", "logValue", "setRotation"):
        assert must in seen, must
    assert os.path.exists(os.path.join(SCALA_DIR, "GpuContext.scala"))


def test_accept_all_evaluator(pkg):
    """api/sampling/evaluators/AcceptAllEvaluator.scala:22-28: the constant 0.0."""
    import numpy as np
    assert pkg.AcceptAllEvaluator().logValue(np.zeros(61)) == 0.0
