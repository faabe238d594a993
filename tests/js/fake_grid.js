// A recording stand-in for ColumnGrid: logs every call the flow driver makes.  Used to compare
// the call sequence of fem_amd/js/runtime.js with the one the reference's runtime produced
// (tests/golden/runtime_trace.json, written by oracle/gen_golden.js with this same class).
'use strict';
const PROBES = [[1, 1], [37, 53], [101, 77], [150.5, 20.25], [199, 199], [60, 141]];

function makeFakeGridClass(log) {
  return class FakeGrid {
    constructor(LX, LY, dx, dy) {
      this.nMaxCache = 10; this.colsCache = {}; this.sliderCache = {};
      log.push(['new', LX, LY, dx, dy]);
    }
    _m(fn) { return fn ? PROBES.map(p => (fn(p[0], p[1]) ? 1 : 0)).join('') : 'null'; }
    deposit_general(fn, mat, t, S) { log.push(['deposit_general', this._m(fn), mat, t, S]); }
    etch_general(fn, mat, t, S) { log.push(['etch_general', this._m(fn), mat, t, S]); }
    deposit_ALD(mat, t, o) { log.push(['deposit_ALD', mat, t, !!(o && o.isCache)]); }
    etch_wet(mat, t, o) { log.push(['etch_wet', mat, t, !!(o && o.isCache)]); }
    strip_connected(mat) { log.push(['strip_connected', mat]); }
    cmp(t, mat) { log.push(['cmp', t, mat]); }
    createNewGrid() { log.push(['createNewGrid']); }
    clearAll() { log.push(['clearAll']); }
    initializeCache() { this.colsCache = {}; this.sliderCache = {}; log.push(['initializeCache']); }
    saveCache(id) {
      while (Object.keys(this.colsCache).length >= this.nMaxCache) {
        const c = Object.entries(this.colsCache).map(([key, v]) => ({ key, num: v[1] }));
        const m = c.reduce((a, b) => (a.num < b.num ? a : b));
        delete this.colsCache[m.key];
      }
      this.colsCache[id] = [{}, 0];
      log.push(['saveCache', id]);
    }
    loadCache(id) {
      log.push(['loadCache', id === undefined ? null : id]);
      if (!id) return;
      this.colsCache[id][1] += 1;
    }
    clearCacheExcept(ids) {
      Object.keys(this.colsCache).forEach(k => { if (!ids.includes(k)) delete this.colsCache[k]; });
      log.push(['clearCacheExcept', ids.slice()]);
    }
  };
}

// a scripted editing session: (snapshot, opts) pairs
function session() {
  const rect = (l, t, w, h, o) => Object.assign({ type: 'rect', left: l, top: t, width: w, height: h, angle: 0, scaleX: 1, scaleY: 1, visible: true, data: { polarity: 'positive' } }, o || {});
  const materialColor = { Si: { color: '#1', id: 3 }, Ox: { color: '#2', id: 4 }, Nit: { color: '#3', id: 5 }, Poly: { color: '#4', id: 6 } };
  const maskList = [
    { id: 'mask0000000001', name: 'A', data: { objects: [rect(20, 20, 60, 90), rect(90, 10, 40, 40, { angle: 25, data: { polarity: 'negative' } })] } },
    { id: 'mask0000000002', name: 'B', data: { objects: [{ type: 'circle', left: 30, top: 40, radius: 35, width: 70, height: 70, visible: true, angle: 0, scaleX: 1, scaleY: 1 },
      { type: 'polygon', left: 100, top: 100, points: [{ x: 0, y: 0 }, { x: 60, y: 10 }, { x: 20, y: 70 }], visible: true, angle: 0, scaleX: 1.5, scaleY: 1 }] } },
  ];
  const P = (id, kind, material, thickness, mask, extra) => Object.assign({ id, kind, material, thickness, mask: mask || '-', name: id }, extra || {});
  const procs = [
    P('s1', 'SUBSTR', 'Si', 30), { id: 'g1', kind: 'GROUPSTART', unfolded: true, name: 'grp' },
    P('s2', 'DEPO', 'Ox', 20, '-', { conformality: 0 }), P('s3', 'DEPO', 'Nit', 10, '-', { conformality: 1 }),
    P('s4', 'ETCH', 'Nit', 10, 'mask0000000001', { conformality: 0 }), { id: 'g1e', kind: 'GROUPEND', name: 'grp' },
    P('s5', 'ETCH', 'Ox', 12, 'mask0000000001', { conformality: 0.5 }), P('s6', 'NEW', '-', 0),
    P('s7', 'DEPO', 'Poly', 8, 'mask0000000002', { conformality: 0.7 }), P('s8', 'CMP', 'Nit', 6), P('s9', 'STRIP', 'Ox', 1),
    P('s10', 'ETCH', 'ALL', 5, 'deleted'), P('s11', 'ALD', 'Ox', 4), P('s12', 'WETETCH', 'ALL', 3), P('s13', 'DEPO', '-', 5),
    P('s14', 'ETCH', 'ALL', 7, 'mask0000000002', { conformality: 'x' }), P('s15', 'DEPO', 'Si', 0), P('s16', 'DEPO', 'Si', 9),
  ];
  const snap = (arrow, opts, mutate) => {
    const processes = JSON.parse(JSON.stringify(procs));
    if (mutate) mutate(processes);
    return [{ processes, arrowBoundId: arrow, materialColor, opts }, opts];
  };
  const set = (id, key, val) => ps => { ps.find(p => p.id === id)[key] = val; };
  const events = [
    snap('s5', { typ: 'process', procId: null }),
    snap('s5', { typ: 'process', procId: null }),                        // unchanged: early return
    snap('s10', { typ: 'explorer', procId: null }),
    snap('s16', { typ: 'explorer', procId: null }),
    snap('s16', { typ: 'inspector', procId: 's7' }, set('s7', 'thickness', 11)),
    snap('s16', { typ: 'sliderdown', procId: 's9' }, set('s7', 'thickness', 11)),
    snap('s16', { typ: 'slidermove', procId: 's9' }, ps => { set('s7', 'thickness', 11)(ps); set('s9', 'thickness', 2)(ps); }),
    snap('s16', { typ: 'slidermove', procId: 's11' }, ps => { set('s7', 'thickness', 11)(ps); set('s11', 'thickness', 6)(ps); }),
    snap('s16', { typ: 'sliderup', procId: 's11' }, ps => { set('s7', 'thickness', 11)(ps); set('s11', 'thickness', 6)(ps); }),
    snap('s16', { typ: 'maskchange', procId: 's4' }, set('s7', 'thickness', 11)),
    snap('s3', { typ: 'explorer', procId: null }, set('s7', 'thickness', 11)),
    snap('s16', { typ: 'process', procId: 's2' }, ps => { ps.splice(2, 1); }),
    snap(null, { typ: 'process', procId: null }),
  ];
  return { materialColor, maskList, events };
}

module.exports = { makeFakeGridClass, session, PROBES };
