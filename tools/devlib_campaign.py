"""The device library's HOST build over seeds the suite does not use (CPU only): the seed-parametrised tests of
tests/test_device_lib_on_host.py, tests/test_decimal.py (oracle side) and tests/test_registry_tail_r3.py (host side), called directly.
      python tools/devlib_campaign.py <first_seed> <last_seed>"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_device_lib_on_host as H
import test_decimal as D
import test_registry_tail_r3 as R3
lib = C.CDLL(H.LIB)
lo, hi = int(sys.argv[1]), int(sys.argv[2])
cases = [("decimal functions", lambda s: H.test_device_decimal_functions_are_exact_on_dense_digits(lib, s)),
         ("decimal casts / compares", lambda s: H.test_device_decimal_casts_and_compares_on_dense_digits(lib, s)),
         ("string functions", lambda s: H.test_device_string_functions_match_oracle_on_host(lib, s)),
         ("like", lambda s: H.test_device_general_like_matcher_on_host(lib, s)),
         ("round / float casts", lambda s: H.test_device_round_and_float_casts_on_host(lib, s)),
         ("oracle decimal ops", D.test_oracle_decimal_ops_on_dense_random_digits),
         ("oracle decimal rounding", D.test_oracle_decimal_rounding_matches_python_decimal),
         ("month differences", lambda s: R3.test_device_month_differences_on_host(lib, s)),
         ("reverse / pad / integer text", lambda s: R3.test_device_reverse_pad_and_integer_text_on_host(lib, s)),
         ("locate / character positions", lambda s: R3.test_device_word_at_a_time_locate_and_character_positions_on_host(lib, s))]
bad = 0
for seed in range(lo, hi):
    for name, fn in cases:
        try:
            fn(seed)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print("FAIL", name, seed, type(e).__name__, str(e)[:300].replace("\n", " "), flush=True)
print(f"seeds {lo}..{hi - 1} x {len(cases)} host / oracle checks: {bad} failures")
