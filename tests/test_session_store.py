"""Evolving-session store (srn_session_store_*, srn_session_key): the reference's RocksDBSessionStore semantics
(src/sessions/mod.rs:37-72) and its session key (MD5 digest as u128, recommend_resource.rs:27-28).  Host code only."""
import hashlib
import threading

import pytest


def test_session_key_is_the_md5_digest_big_endian():
    from serenade_amd.serving import session_key
    for s in ["", "a", "abc", "message digest", "x" * 55, "y" * 56, "z" * 63, "z" * 64, "w" * 119, "q" * 120, "q" * 1000, "sessie-éü"]:
        assert session_key(s) == int.from_bytes(hashlib.md5(s.encode()).digest(), "big"), s
    assert session_key("") == 0xd41d8cd98f00b204e9800998ecf8427e          # RFC 1321 test suite
    assert session_key("abc") == 0x900150983cd24fb0d6963f7d28e17f72


def test_idle_and_ttl_clocks():
    from serenade_amd.serving import SessionStore, session_key
    st = SessionStore()                                  # reference defaults: ttl 30 min (serving.rs:55), idle 20 min (mod.rs:35)
    k = session_key("abc")
    assert st.get_session_items(k, now=1000) == []       # unknown -> empty (mod.rs:55)
    st.update_session_items(k, [7, 8, 9], now=1000)
    assert st.get_session_items(k, now=1000) == [7, 8, 9]
    assert st.get_session_items(k, now=1000 + 1200) == [7, 8, 9]   # "<=" in mod.rs:49
    assert st.get_session_items(k, now=1000 + 1201) == []
    st.update_session_items(k, [9], now=2300)            # an update restarts the clock and replaces the items
    assert st.get_session_items(k, now=3400) == [9]
    assert st.sweep(now=2300 + 1800) == 1
    assert st.sweep(now=2300 + 1801) == 0
    assert st.get_session_items(k, now=2300 + 1801) == []
    st.update_session_items(k, [], now=5000)             # an empty session is a valid value
    assert st.get_session_items(k, now=5000) == []
    from serenade_amd import SerenadeError
    st.update_session_items(k, list(range(40)), now=5000)
    with pytest.raises(SerenadeError):
        st.get_session_items(k, now=5000, cap=8)
    st.close()


def test_concurrent_updates_keep_sessions_apart():
    from serenade_amd.serving import SessionStore, session_key
    st = SessionStore()
    keys = [session_key("s%d" % i) for i in range(2000)]
    errs = []

    def worker(t):
        try:
            for r in range(3):
                for i in range(t, len(keys), 8):
                    st.update_session_items(keys[i], [i, r, t], now=100 + r)
            for i in range(t, len(keys), 8):
                assert st.get_session_items(keys[i], now=110) == [i, 2, t]
        except Exception as e:   # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    assert st.sweep(now=110) == len(keys)
    st.close()
