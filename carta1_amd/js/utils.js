// pipe() and throwError(): the composition helpers the reference exports (codec/utils.js:14-30).
export function throwError(msg) {
  throw new Error(msg)
}

export function pipe(context, ...stages) {
  const fns = stages.map((stage) => stage(context))
  return (input) => fns.reduce((value, fn) => fn(value), input)
}
