// Headless flow driver with the behaviour of the reference's ProcessRuntime
// (template/script_runtime.js:6-341): turns a 'simflow:changed' snapshot into grid calls, with
// the same step dispatch, the same mask predicate and the same replay-from-cache policy.
//
// The reference reads its collaborators from browser globals (window.prj.gridRenderer,
// .ColumnGrid, .processFlow.materialColor, .maskmanager.maskList).  Here they are passed in,
// so the driver runs under Node; `attachToWindow` wires the same object to those globals and
// to the 'simflow:changed' event when a window-like object exists.
'use strict';
const { ColumnGrid, maskFunctionFor } = require('./columngrid.js');

const STEP_KINDS_NEEDING_MATERIAL = ['DEPO', 'ALD', 'ETCH', 'WETETCH', 'STRIP'];

// ---- the mask predicate, host form (script_runtime.js:54-141).  true = the step applies here.
function pointInPolygon(x, y, points) {
  let inside = false;
  for (let a = 0, b = points.length - 1; a < points.length; b = a++) {
    const xa = points[a].x, ya = points[a].y, xb = points[b].x, yb = points[b].y;
    if (((ya > y) !== (yb > y)) && (x < (xb - xa) * (y - ya) / (yb - ya + 1e-9) + xa)) inside = !inside;
  }
  return inside;
}

function pointInsideShape(x, y, obj) {
  const w = obj.width || 0, h = obj.height || 0, angle = obj.angle || 0;
  const ddx = x - (obj.left || 0), ddy = y - (obj.top || 0);
  let lx = ddx, ly = ddy;
  if (angle !== 0) {
    const rad = (-angle * Math.PI) / 180;
    lx = ddx * Math.cos(rad) - ddy * Math.sin(rad);
    ly = ddx * Math.sin(rad) + ddy * Math.cos(rad);
  }
  lx /= (obj.scaleX || 1);
  ly /= (obj.scaleY || 1);
  switch (obj.type) {
    case 'rect': return lx >= 0 && lx <= w && ly >= 0 && ly <= h;
    case 'circle': { const r = obj.radius || w / 2; return (lx - w / 2) ** 2 + (ly - h / 2) ** 2 <= r ** 2; }
    case 'ellipse': {
      const rx = obj.rx || w / 2, ry = obj.ry || h / 2, ex = lx - w / 2, ey = ly - h / 2;
      return (ex * ex) / (rx * rx) + (ey * ey) / (ry * ry) <= 1;
    }
    case 'polygon': return pointInPolygon(lx, ly, obj.points);
    case 'path':
      if (!obj.path) return false;
      return pointInPolygon(lx, ly, obj.path.filter(p => p[0] === 'L' || p[0] === 'M').map(p => ({ x: p[1], y: p[2] })));
    default: return false;
  }
}

function isPointOpen(x, y, maskData) {
  if (!maskData || !maskData.objects) return true;
  for (let k = maskData.objects.length - 1; k >= 0; k--) {
    const obj = maskData.objects[k];
    if (!obj.visible) continue;
    if (pointInsideShape(x, y, obj)) return ((obj.data && obj.data.polarity) || 'positive') !== 'positive';
  }
  return true;
}

function deepEqual(a, b) {                                   // script_runtime.js:326-340
  if (a === b) return true;
  if (typeof a !== 'object' || typeof b !== 'object' || a === null || b === null) return false;
  const ka = Object.keys(a), kb = Object.keys(b);
  if (ka.length !== kb.length) return false;
  for (const k of ka) { if (!kb.includes(k)) return false; if (!deepEqual(a[k], b[k])) return false; }
  return true;
}

class ProcessRuntime {
  // env: { domain?, ColumnGrid?, gridRenderer?, materialColor() or processFlow, maskList() or maskmanager }
  constructor(env = {}) {
    this.env = env;
    this.renderer3D = env.gridRenderer || { updateFromGrid() {} };
    this.domain = env.domain || { LX: 200, LY: 200, dx: 2, dy: 2 };           // script_runtime.js:15
    const Grid = env.ColumnGrid || ColumnGrid;
    const { LX, LY, dx, dy } = this.domain;
    // env.devices (no counterpart in the reference): the HIP devices to cut the grid over, one slab each
    this.grid = env.devices ? new Grid(LX, LY, dx, dy, 32, 0.1, { devices: env.devices }) : new Grid(LX, LY, dx, dy);
    this.oldUpto = null;
    this._maskFns = new Map();         // mask id -> { data, fn }: the closure made for that mask record's data
  }

  _materialColor() { return this.env.processFlow ? this.env.processFlow.materialColor : (this.env.materialColor || {}); }
  _maskList() { return this.env.maskmanager ? this.env.maskmanager.maskList : (this.env.maskList || []); }

  _arrowGapIndex(processes, boundId) {                       // script_runtime.js:39-43
    if (!boundId) return 0;
    const idx = processes.findIndex(p => p.id === boundId);
    return idx < 0 ? 0 : idx + 1;
  }

  // script_runtime.js:45-52.  The closure answers like the reference's and also carries the
  // reduced shapes, so the GPU grid rasterises it itself instead of calling it per cell.
  // One closure per mask record is kept for as long as the record holds the same `data` object: the mask editor
  // replaces `data` by a fresh canvas.toJSON() whenever a mask is saved (template/script_maskeditor.js:206, 240-245),
  // so identity of `data` is the record's version.  A flow of many steps through one mask then reduces its shapes
  // once, and the grid (which keys its device rasters by the reduced list) rasterises it once.
  _getMaskFun(maskid) {
    if (!maskid || maskid == '-') return () => true;
    const rec = this._maskList().find(m => m.id === maskid);
    const data = rec.data;                                    // a missing record throws, as in the reference
    const kept = this._maskFns.get(maskid);
    if (kept && kept.data === data) return kept.fn;
    const fn = maskFunctionFor(data, (x, y) => isPointOpen(x, y, data));
    this._maskFns.set(maskid, { data, fn });
    return fn;
  }

  _applyStep(grid, step, useProcCache = false) {              // script_runtime.js:143-181
    if (step.kind === 'GROUPSTART' || step.kind === 'GROUPEND') return;
    if (step.kind === 'NEW') return;
    if (STEP_KINDS_NEEDING_MATERIAL.includes(step.kind) && (step.material === '-' || step.material === '')) return;
    if (step.mask === 'deleted') return;
    const kind = (step.kind || '').toUpperCase();
    const mat = step.material == 'ALL' ? 255
      : (step.material == '' || step.material == '-') ? 0
      : (this._materialColor()[step.material].id || null);
    const thk = Number(step.thickness || 0);
    const conformality = (typeof step.conformality === 'number') ? step.conformality : 0;
    const maskfun = this._getMaskFun(step.mask);
    if (thk <= 0) return;
    switch (kind) {
      case 'SUBSTR': grid.deposit_general(maskfun, mat, thk, 0); break;
      case 'DEPO': grid.deposit_general(maskfun, mat, thk, conformality); break;
      case 'ALD': grid.deposit_ALD(mat, thk, { isCache: useProcCache }); break;
      case 'ETCH': grid.etch_general(maskfun, mat, thk, conformality); break;
      case 'WETETCH': grid.etch_wet(mat, thk, { isCache: useProcCache }); break;
      case 'STRIP': grid.strip_connected(mat); break;
      case 'CMP': grid.cmp(thk, mat); break;
      default: break;
    }
  }

  // script_runtime.js:183-299
  _build(snapshot, opts) {
    const grid = this.grid;
    const all = (snapshot && snapshot.processes) || [];
    const processes = all.filter(p => p.kind !== 'GROUPSTART' && p.kind !== 'GROUPEND');
    const saveEvery = Math.max(3, Math.floor(processes.length / grid.nMaxCache));
    const upto = this._arrowGapIndex(processes, snapshot && snapshot.arrowBoundId);
    const stepsUptoNow = processes.slice(0, upto);
    const changed = this._arrowGapIndex(processes, opts.procId);
    const typ = opts.typ;
    const searchFrom = typ === 'explorer' ? upto - 1 : typ === 'process' ? changed - 1 : changed - 2;

    let cacheId = null, startAt = 0;
    const cachedIds = [];
    for (let p = searchFrom; p >= 0; p--) {
      if (processes[p].id in grid.colsCache) {
        if (!cacheId) { cacheId = processes[p].id; startAt = p + 1; }
        cachedIds.push(processes[p].id);
      }
    }

    if (typ === 'process' && deepEqual(this.oldUpto, stepsUptoNow)) return;
    if ((typ === 'process' || typ === 'maskchange') && cacheId === null) {
      grid.initializeCache(processes[0].id);
      grid.createNewGrid();
    } else {
      grid.loadCache(cacheId);
      if (typ !== 'explorer') grid.clearCacheExcept(cachedIds);
    }

    const isSlow = s => s.kind === 'ALD' || s.kind === 'WETETCH';
    const replay = (after) => {
      let sinceSave = 0;
      for (let n = startAt; n < upto; n++) {
        const step = processes[n];
        sinceSave += 1;
        this._applyStep(grid, step, typ === 'slidermove' ? (changed === n + 1) : false);
        sinceSave = after(n, step, sinceSave);
      }
    };

    if (typ === 'process' || typ === 'maskchange') {
      replay((n, step, c) => { if (c === saveEvery || isSlow(step)) { grid.saveCache(step.id); return 0; } return c; });
    } else if (typ === 'explorer') {
      if (deepEqual(this.oldUpto, stepsUptoNow)) return;
      replay((n, step, c) => {
        if (step.id in grid.colsCache) c = 0;
        if (c === saveEvery || isSlow(step)) { grid.saveCache(step.id); return 0; }
        return c;
      });
      if (cacheId) grid.colsCache[cacheId][1] += 1;
    } else if (typ === 'inspector' || typ == 'sliderup') {
      replay((n, step, c) => { if (n === changed - 2 || c === saveEvery || isSlow(step)) { grid.saveCache(step.id); return 0; } return c; });
      if (deepEqual(this.oldUpto, stepsUptoNow)) return;
      if (cacheId) grid.colsCache[cacheId][1] += 1;
    } else if (typ === 'sliderdown') {
      if (!grid.sliderCache.changedProcIndex || grid.sliderCache.changedProcIndex !== changed) grid.sliderCache = { changedProcIndex: changed };
      replay((n, step, c) => { if (n === changed - 2) grid.saveCache(step.id); return c; });
      if (deepEqual(this.oldUpto, stepsUptoNow)) return;
    } else if (typ === 'slidermove') {
      replay((n, step, c) => c);
    }

    this.oldUpto = stepsUptoNow;
    this.renderer3D.updateFromGrid(grid, (snapshot && snapshot.materialColor) || {});
  }

  // entry point of the reference: the 'simflow:changed' event (script_runtime.js:26-30)
  onSimflowChanged(detail) {
    const opts = (detail && detail.opts) ? detail.opts : { typ: 'process', procId: null };
    this._build(detail, opts);
  }
}

// Wire a runtime to browser-style globals: window.prj.{gridRenderer, ColumnGrid, processFlow,
// maskmanager} and the 'simflow:changed' event.
function attachToWindow(win, domain) {
  win.prj = win.prj || {};
  win.prj.ColumnGrid = win.prj.ColumnGrid || ColumnGrid;
  const rt = new ProcessRuntime({ domain, ColumnGrid: win.prj.ColumnGrid, gridRenderer: win.prj.gridRenderer,
    get processFlow() { return win.prj.processFlow; }, get maskmanager() { return win.prj.maskmanager; } });
  win.addEventListener('simflow:changed', ev => rt.onSimflowChanged(ev.detail));
  win.prj.processRuntime = rt;
  return rt;
}

module.exports = { ProcessRuntime, attachToWindow, isPointOpen, pointInsideShape, pointInPolygon, deepEqual };
