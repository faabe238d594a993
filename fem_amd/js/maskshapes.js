// Reduce a mask record's fabric.js JSON to the plain numbers the device rasteriser takes.
//
// The reference decides "does the step apply at (x, y)" by walking maskData.objects from the
// last to the first and letting the first containing shape decide
// (template/script_runtime.js:54-71); each shape test starts with
//   ox = obj.left||0, oy = obj.top||0, w = obj.width||0, h = obj.height||0,
//   angle = obj.angle||0, scaleX = obj.scaleX||1, scaleY = obj.scaleY||1
// and rotates by rad = (-angle*PI)/180 when angle !== 0 (script_runtime.js:73-95).
// Everything that does not depend on (x, y) is evaluated here, once per shape, with this
// JavaScript engine's own Math.cos / Math.sin, so the device only does +,-,*,/ and compares.
'use strict';

const SHAPE_RECT = 0, SHAPE_CIRCLE = 1, SHAPE_ELLIPSE = 2, SHAPE_POLYGON = 3, SHAPE_NEVER = 4;
const SHAPE_BYTES = 104;          // sizeof(cg_shape): 6 x int32 + 10 x double

// -> { open: true }                         no objects list: every column is open (rt:56)
//    { shapes: [[...15 numbers]], pts: [] } otherwise; invisible objects are dropped (rt:62)
function reduceMask(maskData) {
  reduceMask.calls += 1;                   // (how often a host walks a mask's objects: the tests watch it)
  if (!maskData || !maskData.objects) return { open: true, shapes: [], pts: [] };
  const shapes = [], pts = [];
  for (const obj of maskData.objects) {
    if (!obj.visible) continue;
    const left = obj.left || 0, top = obj.top || 0;
    const w = obj.width || 0, h = obj.height || 0;
    const angle = obj.angle || 0;
    const scaleX = obj.scaleX || 1, scaleY = obj.scaleY || 1;
    const polarity = (obj.data && obj.data.polarity) || 'positive';
    const positive = polarity === 'positive' ? 1 : 0;
    let cosr = 1, sinr = 0;
    const rotated = angle !== 0 ? 1 : 0;
    if (rotated) {
      const rad = (-angle * Math.PI) / 180;
      cosr = Math.cos(rad); sinr = Math.sin(rad);
    }
    let type = SHAPE_NEVER, a = 0, b = 0, ptOff = 0, ptCnt = 0;
    switch (obj.type) {
      case 'rect': type = SHAPE_RECT; break;
      case 'circle': type = SHAPE_CIRCLE; a = obj.radius || w / 2; break;
      case 'ellipse': type = SHAPE_ELLIPSE; a = obj.rx || w / 2; b = obj.ry || h / 2; break;
      case 'polygon': {
        if (!obj.points) throw new TypeError('polygon without points');   // the reference throws here too (rt:133)
        type = SHAPE_POLYGON; ptOff = pts.length / 2; ptCnt = obj.points.length;
        for (const p of obj.points) pts.push(p.x, p.y);
        break;
      }
      case 'path': {
        if (!obj.path) break;                                             // rt:120 -> never inside
        type = SHAPE_POLYGON; ptOff = pts.length / 2;
        for (const p of obj.path) if (p[0] === 'L' || p[0] === 'M') { pts.push(p[1], p[2]); ptCnt++; }
        break;
      }
      default: break;
    }
    shapes.push([type, positive, rotated, ptOff, ptCnt, left, top, w, h, cosr, sinr, scaleX, scaleY, a, b]);
  }
  return { open: false, shapes, pts };
}

// Pack reduceMask() output into the byte layout of `cg_shape[]` and a Float64Array of points.
function packShapes(reduced) {
  const n = reduced.shapes.length;
  const buf = new ArrayBuffer(Math.max(1, n) * SHAPE_BYTES);
  const dv = new DataView(buf);
  for (let s = 0; s < n; s++) {
    const r = reduced.shapes[s], o = s * SHAPE_BYTES;
    for (let q = 0; q < 5; q++) dv.setInt32(o + 4 * q, r[q], true);
    dv.setInt32(o + 20, 0, true);
    for (let q = 0; q < 10; q++) dv.setFloat64(o + 24 + 8 * q, r[5 + q], true);
  }
  return { shapes: new Uint8Array(buf), nshapes: n, pts: Float64Array.from(reduced.pts) };
}

reduceMask.calls = 0;

module.exports = { reduceMask, packShapes, SHAPE_BYTES,
  SHAPE_RECT, SHAPE_CIRCLE, SHAPE_ELLIPSE, SHAPE_POLYGON, SHAPE_NEVER };
